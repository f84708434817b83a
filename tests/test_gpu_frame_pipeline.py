"""GPU: the frame pipeline.  Render calls of few paths (one progressive frame per call, the reference's
main loop: src/main.cpp:163-202, RayEngine::render, src/RayEngine.cpp:28-42) are queued on several
streams in turn so that consecutive frames overlap; their resolve passes are chained, so the sum
buffer must hold exactly what the same calls give one after the other, and what one batched call of
the same frames gives."""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


def _backend(scene, W, H, cam, rtype, pipeline):
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_PIPELINE, 1 if pipeline else 0)
    b.upload_scene(scene)
    b.resize(W, H)
    b.set_frame(cam, scene.sky, rtype)
    return b


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", [("cornell_knot", dict(nu=256, nv=32)), ("textured_room", dict(nu=64, nv=16, tex_size=32)), ("ties", dict(nu=24, nv=7, seed=9))])
def test_overlapping_frames_give_the_same_sum(name, kw, rtype):
    W, H, frames = 320, 200, 24
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=1)
    images = {}
    for pipeline in (True, False):
        b = _backend(scene, W, H, cam, rtype, pipeline)
        try:
            for f in range(frames):
                b.render(1 + f, 1, 1)     # no synchronisation in between: the calls overlap when the pipeline is on
            images[pipeline], n = b.readback()
            assert n == frames
        finally:
            b.close()
    b = _backend(scene, W, H, cam, rtype, True)
    try:
        b.render(1, frames, 1)
        batched, _ = b.readback()
    finally:
        b.close()
    assert np.array_equal(images[True].view(np.uint32), images[False].view(np.uint32))
    assert np.array_equal(images[True].view(np.uint32), batched.view(np.uint32))
    want, _ = OracleScene(scene).render(cam, rtype, 1, 3, 1)
    b = _backend(scene, W, H, cam, rtype, True)
    try:
        for f in range(3):
            b.render(1 + f, 1, 1)
        got, _ = b.readback()
    finally:
        b.close()
    assert np.array_equal(got, want, equal_nan=True)


def test_pipeline_with_resets_display_and_scene_changes():
    """Operations on the sum buffer between pipelined calls keep their place in the order of calls."""
    W, H = 256, 160
    scene = capi.Scene("cornell_knot", nu=128, nv=24)
    other = capi.Scene("mixed", nu=96, nv=16)
    cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=1)
    results = {}
    for pipeline in (True, False):
        b = _backend(scene, W, H, cam, capi.RT_TYPE_RGB, pipeline)
        try:
            out = []
            for f in range(5):
                b.render(1 + f, 1, 1)
            b.reset_accum()                       # discards the five frames, whatever stream they are on
            for f in range(4):
                b.render(10 + f, 1, 1)
            out.append(b.display_rgba8())         # sees all four
            for f in range(3):
                b.render(20 + f, 1, 1)
            out.append(b.readback()[0])
            b.upload_scene(other)                 # a commit waits for every stream
            b.set_frame(cam, other.sky, capi.RT_TYPE_RGB)
            b.reset_accum()
            for f in range(4):
                b.render(1 + f, 1, 1)
            b.render(5, 40, 1)                    # a larger call between small ones
            b.render(45, 1, 1)
            out.append(b.readback()[0])
            results[pipeline] = out
        finally:
            b.close()
    for a, c in zip(results[True], results[False]):
        assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(c).view(np.uint8))


def test_reserve_prepares_every_stream_of_the_pipeline():
    """rt_reserve for a call size that the pipeline rotates over its streams sizes the buffers of all of
    them; the frames that follow give the same image as without the reservation."""
    W, H = 200, 120
    scene = capi.Scene("mixed", nu=64, nv=12)
    cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=1)
    images = []
    for reserve in (True, False):
        b = _backend(scene, W, H, cam, capi.RT_TYPE_RGB, True)
        try:
            if reserve:
                b.reserve(1)
            for f in range(7):
                b.render(1 + f, 1, 1)
            images.append(b.readback()[0])
        finally:
            b.close()
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))


def test_display_without_the_wait_shows_each_frame_as_the_waiting_call_does():
    """rt_display_rgba8_begin/_end after every one-frame call, two and three frames in flight, against
    rt_display_rgba8 after the same frames: every displayed frame and the final sum buffer are the same."""
    import torch

    W, H, frames = 320, 200, 12
    scene = capi.Scene("cornell_knot", nu=256, nv=32)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=1)
    b = _backend(scene, W, H, cam, capi.RT_TYPE_RGB, False)
    try:
        want = []
        for f in range(frames):
            b.render(1 + f, 1, 1)
            want.append(b.display_rgba8())
        want_sum, _ = b.readback()
    finally:
        b.close()
    for in_flight in (2, 3):
        b = _backend(scene, W, H, cam, capi.RT_TYPE_RGB, True)
        try:
            bufs = [torch.empty((H, W, 4), dtype=torch.uint8).pin_memory() for _ in range(in_flight)]
            got = []
            for f in range(frames):
                b.render(1 + f, 1, 1)
                slot = f % in_flight
                b.display_begin(slot, bufs[slot].data_ptr(), bufs[slot].numel())
                if f >= in_flight - 1:
                    done = (f - (in_flight - 1)) % in_flight
                    b.display_end(done)
                    got.append(bufs[done].numpy().copy())
            for f in range(frames - (in_flight - 1), frames):
                b.display_end(f % in_flight)
                got.append(bufs[f % in_flight].numpy().copy())
            got_sum, n = b.readback()
            assert n == frames
        finally:
            b.close()
        assert len(got) == frames
        for f in range(frames):
            assert np.array_equal(got[f], want[f]), f"frame {f}, {in_flight} in flight"
        assert np.array_equal(got_sum.view(np.uint32), want_sum.view(np.uint32))


def test_display_slots_refuse_misuse():
    import ctypes as C

    W, H = 64, 48
    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    cam = capi.make_camera(width=W, height=H, bounces=2, ray_per_pixel=1)
    b = _backend(scene, W, H, cam, capi.RT_TYPE_RGB, True)
    try:
        b.render(1, 1, 1)
        out = np.empty((H, W, 4), dtype=np.uint8)
        ptr, n = out.ctypes.data_as(C.c_void_p), out.size
        assert capi.hip.rt_display_rgba8_end(b._ctx, 0) == capi.RT_ERR_STATE        # nothing begun
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 4, ptr, n) == capi.RT_ERR_INVALID   # no such slot
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 0, ptr, 16) == capi.RT_ERR_INVALID  # wrong size
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 0, ptr, n) == capi.RT_OK          # pageable memory works, it only waits
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 0, ptr, n) == capi.RT_ERR_STATE   # begun twice
        assert capi.hip.rt_display_rgba8_end(b._ctx, 0) == capi.RT_OK
        assert np.array_equal(out, b.display_rgba8())
        import torch
        dev = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")                 # device memory as the destination
        torch.cuda.synchronize()
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 1, C.c_void_p(dev.data_ptr()), dev.numel()) == capi.RT_OK
        assert capi.hip.rt_display_rgba8_end(b._ctx, 1) == capi.RT_OK
        assert np.array_equal(dev.cpu().numpy(), out)
        b.resize(W * 2, H)                                                               # the slot follows the new size
        cam2 = capi.make_camera(width=W * 2, height=H, bounces=2, ray_per_pixel=1)
        b.set_frame(cam2, scene.sky, capi.RT_TYPE_RGB)
        b.render(1, 1, 1)
        out2 = np.empty((H, W * 2, 4), dtype=np.uint8)
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 0, out2.ctypes.data_as(C.c_void_p), out2.size) == capi.RT_OK
        assert capi.hip.rt_display_rgba8_end(b._ctx, 0) == capi.RT_OK
        assert np.array_equal(out2, b.display_rgba8())
        # a frame begun before a resize is complete when the resize returns, and its slot is free: the late end call says so
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 2, out2.ctypes.data_as(C.c_void_p), out2.size) == capi.RT_OK
        b.resize(W, H)
        assert np.array_equal(out2, np.zeros_like(out2)) is False
        assert capi.hip.rt_display_rgba8_end(b._ctx, 2) == capi.RT_ERR_STATE
        b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        b.render(1, 1, 1)
        assert capi.hip.rt_display_rgba8_begin(b._ctx, 2, ptr, n) == capi.RT_OK           # and can be begun at the new size
        assert capi.hip.rt_display_rgba8_end(b._ctx, 2) == capi.RT_OK
    finally:
        b.close()
