"""GPU: the four-wide trees (RT_OPT_WIDE_BVH 1, 2 and 3, the last being the default) against the CPU checker.

The wide tree is derived from the uploaded binary tree (1) or from a tree the library builds itself
over the same triangles (2) and finds every ray's closest hit, but not
in the visiting order of hitModel (src/RayScene.cpp:681-736): among triangles hit at the same
distance the other one may win.  Bars:
  * primary-hit ids equal to the checker's wherever no second triangle lies at the same t
    (a differing pixel must have t equal to the checker's t within 1e-5 relative: a tie);
  * t within 1e-5 relative everywhere (u, v where the ids agree);
  * radiance relative RMSE <= 1e-4 against the checker.
On the scenes below there is no tie at all and everything is equal bit for bit, which is what the
tests report on failure.

Value 3 is value 2 plus an exact answer for ties: a ray that met two triangles of one model at its
closest distance is traced once more in the reference's order.  Its bar is the default path's:
equal to the checker bit for bit, on a scene made of nothing but ties too.
"""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

SCENES = [
    ("cornell_knot", dict(nu=128, nv=24)),
    ("knot_glass", dict(nu=128, nv=24)),
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),
    ("mixed", dict(nu=96, nv=16)),
    ("interior", dict(nu=48, nv=12, tex_size=32)),
    ("fuzz", dict(nu=300, nv=5, seed=21)),
    ("unlit_knot", dict(nu=64, nv=12)),
    ("empty", {}),
]


@pytest.fixture(scope="module", params=[1, 2, 3], ids=["from_uploaded_trees", "library_built_trees", "library_built_trees_exact_ties"])
def wide(request):
    """1: four-wide trees collapsed from the uploaded binary trees; 2: from trees the library builds
    itself; 3: as 2, rays with a same-model tie traced again in the reference's order."""
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_WIDE_BVH, request.param)
    b.wide_mode = request.param
    yield b
    b.close()


def _setup(backend, name, kw, width, height, bounces, rpp, rtype):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=width, height=height, bounces=bounces, ray_per_pixel=rpp)
    backend.upload_scene(scene)
    backend.resize(width, height)
    backend.set_frame(cam, scene.sky, rtype)
    return scene, cam


def rel_rmse(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.mean(np.abs(b)), 1e-12))


def _close(a, b, tol=1e-5):
    return np.abs(a.astype(np.float64) - b) <= tol * np.maximum(np.abs(b.astype(np.float64)), 1e-30)


@pytest.mark.parametrize("name,kw", SCENES)
def test_wide_primary_hits(wide, name, kw):
    W, H = 96, 72
    scene, cam = _setup(wide, name, kw, W, H, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = wide.trace_primary(frame=3)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=3)
    assert _close(gt, ot).all(), f"t differs at {np.argwhere(~_close(gt, ot))[:5]}"
    same = (gm == om) & (gi == oi)
    # a different winner is only acceptable as a distance tie; none is expected on these scenes
    assert same.all(), f"{int((~same).sum())} pixels with another triangle at the same distance: {np.argwhere(~same)[:5]}"
    assert _close(gu[same], ou[same]).all() and _close(gv[same], ov[same]).all()


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", SCENES)
def test_wide_radiance(wide, name, kw, rtype):
    W, H, frames = 64, 48, 3
    scene, cam = _setup(wide, name, kw, W, H, 4, 4, rtype)
    wide.reset_accum()
    wide.render(1, frames, 1)
    got, n = wide.readback()
    want, _ = OracleScene(scene).render(cam, rtype, 1, frames, 1)
    assert n == frames
    finite = np.isfinite(want).all(axis=-1) & np.isfinite(got).all(axis=-1)
    assert np.array_equal(np.isfinite(want), np.isfinite(got))
    err = rel_rmse(got[finite], want[finite]) if finite.any() else 0.0
    differing = int(np.any(got[finite] != want[finite], axis=-1).sum())
    assert err <= 1e-4, f"relative RMSE {err:.3e}, {differing} differing pixels"


def test_wide_deep_stack_spills_to_memory(wide):
    """A dense soup of overlapping triangles makes rays hold more stack entries than the part kept
    on chip; the entries beyond it live in device memory and the result must not change."""
    W, H = 64, 48
    scene, cam = _setup(wide, "fuzz", dict(nu=6000, nv=4, seed=5), W, H, 2, 1, capi.RT_TYPE_RGB)
    wide.set_counters(True)
    try:
        wide.reset_stats()
        wide.reset_accum()
        wide.render(1, 1, 1)
        got, _ = wide.readback()
        deepest = wide.stats()["max_stack"]
    finally:
        wide.set_counters(False)
    if wide.wide_mode == 1:   # the library's own trees separate the soup better and may stay on chip
        assert deepest > 16, f"deepest stack {deepest}: the scene does not reach the in-memory part"
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 1, 1)
    assert rel_rmse(got, want) <= 1e-4


def test_wide_option_rebuilds_committed_scene():
    """Switching the option on a context that already holds a scene re-derives the tree at once."""
    W, H = 64, 48
    b = capi.Backend(0)
    try:
        b.set_option(capi.RT_OPT_WIDE_BVH, 0)
        scene, cam = _setup(b, "cornell_knot", dict(nu=96, nv=16), W, H, 3, 4, capi.RT_TYPE_RGB)
        b.render(1, 2, 1)
        ref, _ = b.readback()
        b.set_option(capi.RT_OPT_WIDE_BVH, 1)
        b.reset_accum()
        b.render(1, 2, 1)
        got, _ = b.readback()
        b.set_option(capi.RT_OPT_WIDE_BVH, 2)
        b.reset_accum()
        b.render(1, 2, 1)
        own, _ = b.readback()
        b.set_option(capi.RT_OPT_WIDE_BVH, 0)
        b.reset_accum()
        b.render(1, 2, 1)
        back, _ = b.readback()
        assert b.set_option(capi.RT_OPT_WIDE_BVH, 0) is None
        with pytest.raises(capi.RtError):
            b.set_option(capi.RT_OPT_WIDE_BVH, 5)
    finally:
        b.close()
    assert rel_rmse(got, ref) <= 1e-4
    assert rel_rmse(own, ref) <= 1e-4
    assert np.array_equal(back, ref)


def test_wide_full_size_image_against_reference_order(wide):
    """Dragon-class mesh at 1080p: the wide tree's primary hits against the binary tree's (both on
    the GPU; the binary tree's are checked against the checker elsewhere)."""
    scene = capi.Scene("cornell_knot", nu=2048, nv=213)
    W, H = 1920, 1080
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=1)
    out = []
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_WIDE_BVH, 0)
    try:
        for be in (b, wide):
            be.upload_scene(scene)
            be.resize(W, H)
            be.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            out.append(be.trace_primary(frame=1))
    finally:
        b.close()
        wide.resize(64, 48)
    (m0, i0, t0, u0, v0), (m1, i1, t1, u1, v1) = out
    assert _close(t1, t0).all()
    differing = (m0 != m1) | (i0 != i1)
    assert differing.mean() <= 1e-5, f"{int(differing.sum())} of {W * H} primary hits picked another triangle"


TIE_SCENES = [
    ("ties", dict(nu=6, nv=4, seed=1)),
    ("ties", dict(nu=24, nv=7, seed=9)),
]


@pytest.mark.parametrize("name,kw", TIE_SCENES)
def test_exact_ties_primary_hits(name, kw):
    """A floor of coplanar copies: every hit on it is a tie between triangles of one model.  The
    four-wide walk alone picks winners of its own; with the re-trace it must report the checker's."""
    W, H = 96, 72
    b = capi.Backend(0)
    try:
        b.set_option(capi.RT_OPT_WIDE_BVH, 3)
        scene, cam = _setup(b, name, kw, W, H, 4, 1, capi.RT_TYPE_RGB)
        gm, gi, gt, gu, gv = b.trace_primary(frame=2)
        b.set_option(capi.RT_OPT_WIDE_BVH, 2)
        lm, li, lt, _, _ = b.trace_primary(frame=2)
    finally:
        b.close()
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=2)
    assert np.array_equal(gm, om) and np.array_equal(gi, oi)
    assert np.array_equal(gt, ot) and np.array_equal(gu, ou) and np.array_equal(gv, ov)
    # the scene does what it is for: without the re-trace some winners differ, at equal distance
    other = (lm != om) | (li != oi)
    assert other.any(), "no tie was resolved differently: the scene does not exercise the re-trace"
    assert np.array_equal(lt, ot)


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", TIE_SCENES)
def test_exact_ties_radiance(name, kw, rtype):
    W, H, frames = 64, 48, 2
    b = capi.Backend(0)
    try:
        b.set_option(capi.RT_OPT_WIDE_BVH, 3)
        scene, cam = _setup(b, name, kw, W, H, 4, 4, rtype)
        b.reset_accum()
        b.render(1, frames, 1)
        got, n = b.readback()
    finally:
        b.close()
    want, _ = OracleScene(scene).render(cam, rtype, 1, frames, 1)
    assert n == frames
    assert np.array_equal(got, want), f"{int(np.any(got != want, axis=-1).sum())} differing pixels"


def test_scene_beyond_the_wide_box_arithmetic_falls_back_to_the_reference_walk():
    """Coordinates beyond 1e15 do not fit the 8-bit child boxes.  Left at its default the library
    then walks the uploaded trees in the reference's order (the checker's result, as always);
    asked by option for four-wide trees of its own it refuses the scene."""
    W, H = 48, 36
    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    info = scene.streams[3].view(np.int32).reshape(-1, 4)
    idents, lights, nodes, location = (int(x) for x in info[1])
    vertex0 = location + lights * 5 + nodes * 9 + idents * 5
    scene.streams[4][vertex0] = 1e18
    cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=1)
    b = capi.Backend(0)
    try:
        b.upload_scene(scene)
        b.resize(W, H)
        b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        b.render(1, 2, 1)
        got, _ = b.readback()
        want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
        assert np.array_equal(got, want, equal_nan=True)
        for mode in (2, 3):              # (1 takes the uploaded boxes as they are)
            with pytest.raises(capi.RtError):
                b.set_option(capi.RT_OPT_WIDE_BVH, mode)   # re-commits a committed scene at once
                b.upload_scene(scene)                      # (a refused commit leaves the context without one)
        b.set_option(capi.RT_OPT_WIDE_BVH, 0)
        b.upload_scene(scene)
        b.reset_accum()
        b.render(1, 2, 1)
        again, _ = b.readback()
        assert np.array_equal(again, want, equal_nan=True)
    finally:
        b.close()


def test_library_trees_do_not_depend_on_the_number_of_build_threads(monkeypatch):
    """Large models are built by several host threads (RT_BUILD_THREADS, read at every commit): the
    tree, and with it every visit count, must be the one a single thread builds."""
    W, H = 96, 64
    scene = capi.Scene("cornell_knot", nu=512, nv=80)     # 81,920 triangles: above the threshold for threads
    cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=1)
    seen = []
    for threads in ("1", "7", "16"):
        monkeypatch.setenv("RT_BUILD_THREADS", threads)
        b = capi.Backend(0)
        try:
            b.upload_scene(scene)
            b.resize(W, H)
            b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            b.set_counters(True)
            b.reset_stats()
            b.render(1, 2, 1)
            img, _ = b.readback()
            st = b.stats()
        finally:
            b.close()
        seen.append((img, st["node_visits"], st["tri_tests"], st["rays"]))
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    for img, nodes, tris, rays in seen:
        assert np.array_equal(img, want)
        assert (nodes, tris, rays) == seen[0][1:]
