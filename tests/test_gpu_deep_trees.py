"""GPU parity on the traversal-kernel instances that small test meshes never reach.

The reference walks a model's tree with `int stack[32]` (src/RayScene.cpp:681-736).  The backend
keeps that stack in LDS and compiles the reference-order kernel for 20, 24 and 32 entries, picked by
the depth of the deepest uploaded tree; the same instances serve as the second walk of rays that met
a distance tie under the default four-wide trees.  A median-split tree over fewer than a million
triangles is at most 20 levels deep, so:
  * lopsided trees (tests/deep_tree.py) over the usual small scenes reach the 24- and 32-entry
    instances, with and without alpha-cut textures, analytic primitives and visit counters, and the
    `ties` scene forces second walks through them;
  * a 4.2 M-triangle mesh (the largest benchmark configuration's class: depth 22) is checked at
    1920x1080 on row bands, in both tree modes, and tile-split.
Bar as everywhere: ids, t, u, v and radiance equal to the CPU checker bit for bit.
"""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene
from deep_tree import with_depth

pytestmark = pytest.mark.gpu

# name -> (shallowest, deepest tree wanted, overlapping): in the overlapping classes every node carries
# the bounds of its whole model, so rays stack one entry per level and the stack really fills up
# (every ray then opens every node: those classes use the small variants of the scenes)
DEPTHS = {"stack24": (21, 24, False), "stack32": (28, 32, False), "stack24_full": (21, 24, True), "stack32_full": (28, 32, True)}
SCENES = [
    ("cornell_knot", dict(nu=128, nv=24)),                    # plain instance
    ("mixed", dict(nu=96, nv=16)),                            # analytic primitives, four models
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),       # alpha-cut texture + primitives
    ("interior", dict(nu=48, nv=12, tex_size=32)),            # alpha-cut texture, six models
    ("ties", dict(nu=24, nv=7, seed=9)),                      # every floor hit is a same-model tie
]


@pytest.fixture(scope="module", params=[3, 0], ids=["default_trees", "reference_order_walk"])
def backend(request):
    b = capi.Backend(0)
    if request.param != 3:
        b.set_option(capi.RT_OPT_WIDE_BVH, request.param)
    b.tree_mode = request.param
    yield b
    b.close()


SMALL = {"cornell_knot": dict(nu=32, nv=8), "mixed": dict(nu=32, nv=8), "textured_room": dict(nu=32, nv=8, tex_size=32),
         "interior": dict(nu=24, nv=6, tex_size=32), "ties": dict(nu=8, nv=4, seed=9)}
_cache = {}


def deep_scene(name, kw, depth_class):
    lo, hi, overlapping = DEPTHS[depth_class]
    if overlapping:
        kw = SMALL[name]
    key = (name, tuple(sorted(kw.items())), depth_class)
    if key not in _cache:
        scene, depths = with_depth(lambda: capi.Scene(name, **kw), lo, hi, overlapping)
        _cache[key] = (scene, OracleScene(scene), max(depths))
    return _cache[key]


def _frame(backend, scene, width, height, bounces, rpp, rtype):
    cam = capi.make_camera(width=width, height=height, bounces=bounces, ray_per_pixel=rpp)
    backend.upload_scene(scene)
    backend.resize(width, height)
    backend.set_frame(cam, scene.sky, rtype)
    return cam


@pytest.mark.parametrize("depth_class", sorted(DEPTHS))
@pytest.mark.parametrize("name,kw", SCENES)
def test_deep_tree_primary_hits(backend, name, kw, depth_class):
    scene, orc, depth = deep_scene(name, kw, depth_class)
    cam = _frame(backend, scene, 96, 72, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = backend.trace_primary(frame=2)
    om, oi, ot, ou, ov, _ = orc.trace_primary(cam, frame=2)
    assert np.array_equal(gm, om) and np.array_equal(gi, oi), f"depth {depth}: ids differ at {np.argwhere((gm != om) | (gi != oi))[:5]}"
    assert np.array_equal(gt, ot) and np.array_equal(gu, ou) and np.array_equal(gv, ov)


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("depth_class", sorted(DEPTHS))
@pytest.mark.parametrize("name,kw", SCENES)
def test_deep_tree_radiance(backend, name, kw, depth_class, rtype):
    scene, orc, depth = deep_scene(name, kw, depth_class)
    cam = _frame(backend, scene, 64, 48, 4, 4, rtype)
    backend.reset_stats()
    backend.render(1, 2, 1)
    got, n = backend.readback()
    st = backend.stats()
    want, ost = orc.render(cam, rtype, 1, 2, 1)
    assert n == 2 and st["rays"] == ost["rays"]
    assert np.array_equal(got, want, equal_nan=True), f"depth {depth}: {int(np.any(got != want, axis=-1).sum())} differing pixels"
    if name == "ties" and backend.tree_mode == 3:
        assert st["retraced"] > 0, "no ray went through the second walk: the scene does not exercise it"


@pytest.mark.parametrize("depth_class", sorted(DEPTHS))
@pytest.mark.parametrize("name,kw", SCENES)
def test_deep_tree_visit_counters_and_stack_use(backend, name, kw, depth_class):
    """The counting instances of the same kernels: same image, the reference algorithm's visit counts
    (reference-order walk, full closest-hit shadow rays), and a traversal stack that really gets
    deeper than the next smaller instance could hold."""
    scene, orc, depth = deep_scene(name, kw, depth_class)
    cam = _frame(backend, scene, 48, 40, 3, 1, capi.RT_TYPE_RGB)
    backend.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 0)
    backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    backend.set_counters(True)
    try:
        backend.reset_stats()
        backend.render(1, 2, 1)
        got, _ = backend.readback()
        st = backend.stats()
    finally:
        backend.set_counters(False)
        backend.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 1)
    want, ost = orc.render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    assert np.array_equal(got, want)
    assert st["rays"] == ost["rays"] and st["shaded_hits"] == ost["shaded_hits"]
    if backend.tree_mode == 0:
        assert st["node_visits"] == ost["node_visits"] and st["tri_tests"] == ost["tri_tests"]
        # entries held at once never exceed depth - 1; with overlapping children a ray gets close to
        # that, beyond what the next smaller instance of the kernel could hold
        assert st["max_stack"] < depth
        if depth_class.endswith("_full"):
            assert st["max_stack"] == depth - 1, f"deepest stack {st['max_stack']} of a {depth}-level tree"


# ---- the largest benchmark configuration's mesh class: 4.2 M triangles, tree depth 22 ----------------
@pytest.fixture(scope="module")
def kitchen_class():
    scene = capi.Scene("cornell_knot", nu=4096, nv=512)
    assert scene.model_stats(1)["triangles"] == 2 * 4096 * 512
    assert scene.model_stats(1)["max_height"] > 20          # beyond the 20-entry instance
    return scene, OracleScene(scene)


def test_4m_triangle_mesh_primary_ids_on_row_bands(backend, kitchen_class):
    scene, orc = kitchen_class
    W, H = 1920, 1080
    cam = _frame(backend, scene, W, H, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = backend.trace_primary(frame=1)
    assert (gm >= 0).all() and (gm == 1).mean() > 0.18
    for y0 in (8, 290, 532, 810, 1060):
        om, oi, ot, ou, ov, _ = orc.trace_primary(cam, frame=1, region=(0, y0, W, y0 + 12))
        sl = slice(y0, y0 + 12)
        assert np.array_equal(gm[sl], om[sl]) and np.array_equal(gi[sl], oi[sl])
        assert np.array_equal(gt[sl], ot[sl]) and np.array_equal(gu[sl], ou[sl]) and np.array_equal(gv[sl], ov[sl])


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
def test_4m_triangle_mesh_radiance_band(backend, kitchen_class, rtype):
    scene, orc = kitchen_class
    W, H = 1920, 1080
    cam = _frame(backend, scene, W, H, 4, 1, rtype)
    backend.render(1, 3, 1)
    got, n = backend.readback()
    assert n == 3 and np.isfinite(got).all()
    y0 = 536
    want, _ = orc.render(cam, rtype, 1, 3, 1, region=(0, y0, W, y0 + 8))
    assert np.array_equal(got[y0:y0 + 8], want[y0:y0 + 8])


def test_4m_triangle_mesh_tile_split(backend, kitchen_class):
    """The tile-sharded configuration: three shards of 8x8 tiles partition the 1080p image, each pixel
    carrying exactly the unsharded render's value, and a band of the sum equals the checker's."""
    scene, orc = kitchen_class
    W, H = 1920, 1080
    cam = _frame(backend, scene, W, H, 4, 1, capi.RT_TYPE_RGB)
    backend.render(1, 2, 1)
    full, _ = backend.readback()
    total = np.zeros_like(full)
    owners = np.zeros((H, W), np.int32)
    try:
        for k in range(3):
            backend.set_tile_shard(k, 3)
            backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            backend.reset_accum()
            backend.render(1, 2, 1)
            part, _ = backend.readback()
            owners += (part[..., 3] > 0)
            total += part
    finally:
        backend.set_tile_shard(0, 1)
        backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    assert (owners == 1).all()
    assert np.array_equal(total, full)
    y0 = 300
    want, _ = orc.render(cam, capi.RT_TYPE_RGB, 1, 2, 1, region=(0, y0, W, y0 + 8))
    assert np.array_equal(total[y0:y0 + 8], want[y0:y0 + 8])


# ---- the default trees against the reference-order walk at full image sizes ----------------------------
FULL_SIZE = [
    ("cornell_knot", dict(nu=2048, nv=213), 1920, 1080, capi.RT_TYPE_RGB),
    ("knot_glass", dict(nu=2048, nv=213), 1920, 1080, capi.RT_TYPE_SPECTRAL),
    ("interior", dict(nu=512, nv=64, tex_size=256), 1920, 1080, capi.RT_TYPE_RGB),
    ("textured_room", dict(nu=256, nv=32, tex_size=128), 1280, 720, capi.RT_TYPE_RGB),
    ("ties", dict(nu=64, nv=5, seed=3), 1280, 720, capi.RT_TYPE_SPECTRAL),
    ("fuzz", dict(nu=20000, nv=9, seed=5), 1280, 720, capi.RT_TYPE_RGB),
]


@pytest.mark.parametrize("name,kw,W,H,rtype", FULL_SIZE)
def test_default_trees_equal_reference_order_walk_at_full_size(name, kw, W, H, rtype):
    """16 spp of a whole image on the library's four-wide trees (ties traced again) and on the plain
    reference-order walk of the uploaded trees: 0 differing pixels; and a band of it equals the checker."""
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=16)
    images, stats = {}, {}
    for mode in (0, 3):
        be = capi.Backend(0)
        try:
            be.set_option(capi.RT_OPT_WIDE_BVH, mode)
            be.upload_scene(scene)
            be.resize(W, H)
            be.set_frame(cam, scene.sky, rtype)
            be.reset_stats()
            be.render(1, 1, 1)
            images[mode], _ = be.readback()
            stats[mode] = be.stats()
        finally:
            be.close()
    differing = int(np.any(images[0].view(np.uint32) != images[3].view(np.uint32), axis=-1).sum())
    print(f"{name} {W}x{H} type={rtype}: {differing} differing pixels, {stats[3]['retraced']} of {stats[3]['rays']} rays traced again")
    assert differing == 0
    assert stats[0]["rays"] == stats[3]["rays"]
    y0 = H // 2 + 4
    want, _ = OracleScene(scene).render(cam, rtype, 1, 1, 1, region=(0, y0, W, y0 + 4))
    assert np.array_equal(images[3][y0:y0 + 4], want[y0:y0 + 4], equal_nan=True)
