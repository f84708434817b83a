"""GPU: rt_display_rgba8_noalloc (fixed staging area, no allocation) against rt_display_rgba8."""

import ctypes as C

import numpy as np
import pytest

from raytracer_amd import capi

pytestmark = pytest.mark.gpu


def _frame(backend, name, kw, width, height, frames):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=width, height=height, bounces=3, ray_per_pixel=4)
    backend.upload_scene(scene)
    backend.resize(width, height)
    backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    backend.render(1, frames, 1)


def test_display_without_allocation_equals_the_allocating_entry_point():
    """Two calls in a row, the second on a smaller frame: the bytes the first call left in the
    shared staging area beyond the second frame must not show up anywhere."""
    b = capi.Backend(0)
    try:
        _frame(b, "roughness_demo", {}, 200, 120, 3)
        want = b.display_rgba8()
        got = b.display_rgba8_noalloc()
        assert np.array_equal(got, want)
        assert got[..., :3].std() > 5          # a real picture, not a constant

        _frame(b, "mixed", dict(nu=48, nv=8), 56, 40, 2)
        want_small = b.display_rgba8()
        got_small = b.display_rgba8_noalloc()
        assert np.array_equal(got_small, want_small)
        assert not np.array_equal(got_small, want[:40, :56])

        # a second context on the same GPU shares the staging area and gets its own frame
        other = capi.Backend(0)
        try:
            _frame(other, "cornell_quads", {}, 64, 48, 1)
            assert np.array_equal(other.display_rgba8_noalloc(), other.display_rgba8())
        finally:
            other.close()
        assert np.array_equal(b.display_rgba8_noalloc(), want_small)
    finally:
        b.close()


def test_frames_beyond_the_staging_area_are_refused():
    b = capi.Backend(0)
    try:
        scene = capi.Scene("empty")
        b.upload_scene(scene)
        b.resize(4096, 2100)                   # 8,601,600 pixels > 8,388,608
        out = np.zeros((2100, 4096, 4), np.uint8)
        rc = capi.hip.rt_display_rgba8_noalloc(b._ctx, out.ctypes.data_as(C.c_void_p), out.size)
        assert rc == -1
        assert b"staging area" in capi.hip.rt_last_hip_error(b._ctx)
        assert capi.hip.rt_display_rgba8_noalloc(b._ctx, out.ctypes.data_as(C.c_void_p), 16) == -1   # wrong size
        assert not out.any()
        b.resize(3840, 2160)                   # the largest documented frame fits
        assert np.array_equal(b.display_rgba8_noalloc(), b.display_rgba8())
    finally:
        b.close()


def test_no_device_memory_is_taken_or_returned_by_the_call():
    """Free device memory as the runtime reports it is the same before and after a series of calls (the
    allocating entry point takes and returns a staging buffer per call; this one must not touch the allocator),
    also on the first call at a new, larger resolution."""
    import torch
    b = capi.Backend(0)
    try:
        _frame(b, "cornell_quads", {}, 64, 48, 1)
        b.display_rgba8_noalloc()                    # warm: pins nothing, but the first kernel launch loads code
        _frame(b, "cornell_quads", {}, 640, 360, 1)  # a resolution this entry point has not seen
        b.sync()
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info(0)
        frames = [b.display_rgba8_noalloc() for _ in range(4)]
        free_after, _ = torch.cuda.mem_get_info(0)
        assert free_after == free_before, f"free device memory moved by {free_before - free_after} bytes"
        assert all(np.array_equal(f, frames[0]) for f in frames)
    finally:
        b.close()
