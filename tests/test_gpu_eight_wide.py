"""GPU: the eight-wide trees (RT_OPT_WIDE_BVH 4) against the CPU checker.

Mode 4 walks trees the library builds itself (binned surface-area splits collapsed to eight children
per node, 80-byte nodes) and, like the default mode 3, traces a ray that met two triangles of one
model at its closest distance a second time in the visiting order of hitModel
(src/RayScene.cpp:681-736).  Its bar is therefore the default path's: ids, t, u, v and radiance equal
to the checker bit for bit, on a scene made of nothing but ties too.  Scenes with alpha-cut textures
walk the four-wide trees under this setting (the eight-wide kernel has no alpha-cut phase) and must
come out the same.
"""

import os

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

SCENES = [
    ("cornell_knot", dict(nu=128, nv=24)),
    ("knot_glass", dict(nu=128, nv=24)),
    ("mixed", dict(nu=96, nv=16)),                         # analytic primitives, four models (root table in LDS)
    ("ties", dict(nu=24, nv=7, seed=9)),                   # every floor hit is a same-model tie
    ("fuzz", dict(nu=300, nv=5, seed=21)),
    ("fuzz", dict(nu=6000, nv=4, seed=5)),                 # dense soup of overlapping triangles
    ("unlit_knot", dict(nu=64, nv=12)),
    ("three_color", {}),                                   # analytic primitives only: nothing to build
    ("empty", {}),
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),    # alpha-cut texture: four-wide trees
    ("interior", dict(nu=48, nv=12, tex_size=32)),
]


@pytest.fixture(scope="module", params=[None, "2"], ids=["stack_in_lds", "stack_mostly_in_memory"])
def eight(request):
    """Second instance: only two stack entries per lane stay on chip, so nearly every push and pop
    goes through the in-memory part of the stack."""
    old = os.environ.get("RT_W8_LDS_STACK")
    if request.param is not None:
        os.environ["RT_W8_LDS_STACK"] = request.param
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_WIDE_BVH, 4)
    b.short_stack = request.param is not None
    yield b
    b.close()
    if old is None:
        os.environ.pop("RT_W8_LDS_STACK", None)
    else:
        os.environ["RT_W8_LDS_STACK"] = old


def _setup(backend, name, kw, width, height, bounces, rpp, rtype):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=width, height=height, bounces=bounces, ray_per_pixel=rpp)
    backend.upload_scene(scene)
    backend.resize(width, height)
    backend.set_frame(cam, scene.sky, rtype)
    return scene, cam


@pytest.mark.parametrize("name,kw", SCENES)
def test_eight_wide_primary_hits(eight, name, kw):
    scene, cam = _setup(eight, name, kw, 96, 72, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = eight.trace_primary(frame=3)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=3)
    assert np.array_equal(gm, om) and np.array_equal(gi, oi), f"ids differ at {np.argwhere((gm != om) | (gi != oi))[:5]}"
    assert np.array_equal(gt, ot) and np.array_equal(gu, ou) and np.array_equal(gv, ov)


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", SCENES)
def test_eight_wide_radiance(eight, name, kw, rtype):
    scene, cam = _setup(eight, name, kw, 64, 48, 4, 4, rtype)
    eight.reset_stats()
    eight.reset_accum()
    eight.render(1, 3, 1)
    got, n = eight.readback()
    st = eight.stats()
    want, ost = OracleScene(scene).render(cam, rtype, 1, 3, 1)
    assert n == 3 and st["rays"] == ost["rays"]
    assert np.array_equal(got, want, equal_nan=True), f"{int(np.any(got != want, axis=-1).sum())} differing pixels"
    if name == "ties":
        assert st["retraced"] > 0, "no ray went through the second walk: the scene does not exercise it"


@pytest.mark.parametrize("name,kw", [SCENES[0], SCENES[2], SCENES[5]])
def test_eight_wide_counting_instances(eight, name, kw):
    """The counting instances of the kernel: same image, and a stack that goes beyond what is kept on chip
    when that part is made short."""
    scene, cam = _setup(eight, name, kw, 64, 48, 3, 1, capi.RT_TYPE_RGB)
    # The checker counts a shaded hit for every shadow ray that hits anything (the reference fetches that
    # hit's material, src/RayScene.cpp:762); with the shadow early-out the library skips the fetch for
    # occluded shadow rays and counts fewer (same image).  Full closest-hit shadow rays make the counts comparable.
    eight.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 0)
    eight.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    eight.set_counters(True)
    try:
        eight.reset_stats()
        eight.reset_accum()
        eight.render(1, 2, 1)
        got, _ = eight.readback()
        st = eight.stats()
    finally:
        eight.set_counters(False)
        eight.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 1)
    want, ost = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    assert np.array_equal(got, want)
    assert st["rays"] == ost["rays"] and st["shaded_hits"] == ost["shaded_hits"]
    assert st["node_visits"] > 0 and st["tri_tests"] > 0
    if eight.short_stack:
        assert st["max_stack"] > 2, f"deepest stack {st['max_stack']}: the in-memory part was not reached"


def test_eight_wide_full_size_equals_reference_order_walk():
    """16 spp of a 1280x720 image on the eight-wide trees and on the plain reference-order walk of the
    uploaded trees: 0 differing pixels; a band of it equals the checker."""
    scene = capi.Scene("cornell_knot", nu=1024, nv=128)
    W, H = 1280, 720
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=16)
    images = {}
    for mode in (0, 4):
        be = capi.Backend(0)
        try:
            be.set_option(capi.RT_OPT_WIDE_BVH, mode)
            be.upload_scene(scene)
            be.resize(W, H)
            be.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            be.render(1, 1, 1)
            images[mode], _ = be.readback()
        finally:
            be.close()
    assert int(np.any(images[0].view(np.uint32) != images[4].view(np.uint32), axis=-1).sum()) == 0
    y0 = H // 2
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 1, 1, region=(0, y0, W, y0 + 4))
    assert np.array_equal(images[4][y0:y0 + 4], want[y0:y0 + 4], equal_nan=True)
