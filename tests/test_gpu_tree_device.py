"""GPU: the models' binary trees built on the device (RT_OPT_TREE_BUILD = RT_TREE_BUILD_DEVICE) against the CPU checker.

The device build orders a model's triangles along a Morton curve and builds a radix tree over that order; the
four-wide (or eight-wide) nodes are then made from it exactly as from the host's binned surface-area tree.  The
tree differs, the image may not: a ray's closest hit does not depend on the tree, and a ray that met two
triangles of one model at its closest distance is traced again in the visiting order of hitModel
(src/RayScene.cpp:681-736) as in the default mode.  Bar: ids, t, u, v and radiance equal to the checker bit for
bit, also on a scene made of nothing but ties, on clouds of equal centres (equal Morton keys) and on models of
one, two and three triangles.
"""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

SCENES = [
    ("cornell_knot", dict(nu=128, nv=24)),
    ("knot_glass", dict(nu=128, nv=24)),
    ("mixed", dict(nu=96, nv=16)),
    ("ties", dict(nu=24, nv=7, seed=9)),
    ("fuzz", dict(nu=300, nv=5, seed=21)),
    ("fuzz", dict(nu=6000, nv=4, seed=5)),
    ("unlit_knot", dict(nu=64, nv=12)),
    ("three_color", {}),
    ("empty", {}),
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),
    ("interior", dict(nu=48, nv=12, tex_size=32)),
    ("fuzz", dict(nu=1, nv=2, seed=3)),                    # models of one and three triangles
    ("fuzz", dict(nu=2, nv=2, seed=4)),                    # of two and four
    ("panel_floor", {}),                                   # one model that is a single leaf of two triangles
]


@pytest.fixture(scope="module", params=[3, 4], ids=["four_wide", "eight_wide"])
def device_trees(request):
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_WIDE_BVH, request.param)
    b.set_option(capi.RT_OPT_TREE_BUILD, capi.RT_TREE_BUILD_DEVICE)
    yield b
    b.close()


def _setup(backend, name, kw, width, height, bounces, rpp, rtype):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=width, height=height, bounces=bounces, ray_per_pixel=rpp)
    backend.upload_scene(scene)
    backend.resize(width, height)
    backend.set_frame(cam, scene.sky, rtype)
    return scene, cam


@pytest.mark.parametrize("name,kw", SCENES)
def test_device_tree_primary_hits(device_trees, name, kw):
    scene, cam = _setup(device_trees, name, kw, 96, 72, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = device_trees.trace_primary(frame=3)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=3)
    assert np.array_equal(gm, om) and np.array_equal(gi, oi), f"ids differ at {np.argwhere((gm != om) | (gi != oi))[:5]}"
    assert np.array_equal(gt, ot) and np.array_equal(gu, ou) and np.array_equal(gv, ov)


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", SCENES)
def test_device_tree_radiance(device_trees, name, kw, rtype):
    scene, cam = _setup(device_trees, name, kw, 64, 48, 4, 4, rtype)
    device_trees.reset_stats()
    device_trees.reset_accum()
    device_trees.render(1, 3, 1)
    got, n = device_trees.readback()
    st = device_trees.stats()
    want, ost = OracleScene(scene).render(cam, rtype, 1, 3, 1)
    assert n == 3 and st["rays"] == ost["rays"]
    assert np.array_equal(got, want, equal_nan=True), f"{int(np.any(got != want, axis=-1).sum())} differing pixels"
    if name == "ties":
        assert st["retraced"] > 0, "no ray went through the second walk: the scene does not exercise it"


def test_device_tree_full_size_equals_host_tree():
    """16 spp of a 1280x720 image with the trees from the device and from the host threads: 0 differing pixels;
    the option may be flipped on a committed scene (the scene is committed again)."""
    scene = capi.Scene("cornell_knot", nu=1024, nv=128)
    W, H = 1280, 720
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=16)
    images = {}
    be = capi.Backend(0)
    try:
        be.upload_scene(scene)
        be.resize(W, H)
        for where in (capi.RT_TREE_BUILD_HOST, capi.RT_TREE_BUILD_DEVICE):
            be.set_option(capi.RT_OPT_TREE_BUILD, where)
            be.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            be.reset_accum()
            be.render(1, 1, 1)
            images[where], _ = be.readback()
    finally:
        be.close()
    assert int(np.any(images[0].view(np.uint32) != images[1].view(np.uint32), axis=-1).sum()) == 0


@pytest.mark.parametrize("where", [capi.RT_TREE_BUILD_HOST, capi.RT_TREE_BUILD_DEVICE], ids=["host", "device"])
@pytest.mark.parametrize("value", [1e18, float("nan")])
def test_device_tree_declines_what_the_host_build_declines(value, where):
    """A coordinate beyond 1e15 (or not a number) does not fit the 8-bit child boxes: left at its default tree
    mode the library walks the uploaded trees in the reference's order, whichever side was to build."""
    W, H = 48, 36
    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    info = scene.streams[3].view(np.int32).reshape(-1, 4)
    idents, lights, nodes, location = (int(x) for x in info[1])
    vertex0 = location + lights * 5 + nodes * 9 + idents * 5
    scene.streams[4][vertex0] = value
    cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=1)
    b = capi.Backend(0)
    try:
        b.set_option(capi.RT_OPT_TREE_BUILD, where)
        b.upload_scene(scene)
        assert b.info(capi.RT_INFO_TREE_MODE) == 0
        b.resize(W, H)
        b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        b.render(1, 2, 1)
        got, _ = b.readback()
        if value == value:   # (what a vertex that is not a number looks like is not defined by the reference's shader)
            want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
            assert np.array_equal(got, want, equal_nan=True)
    finally:
        b.close()


def test_device_tree_is_the_same_tree_every_time():
    """Node numbers come from prefix sums, not from atomics: two commits give the same tree, hence the same
    visit counts (and the counts differ from the host trees', so the option did take effect)."""
    scene = capi.Scene("cornell_knot", nu=512, nv=80)
    cam = capi.make_camera(width=96, height=64, bounces=3, ray_per_pixel=1)
    seen = []
    for where in (capi.RT_TREE_BUILD_DEVICE, capi.RT_TREE_BUILD_DEVICE, capi.RT_TREE_BUILD_HOST):
        b = capi.Backend(0)
        try:
            b.set_option(capi.RT_OPT_TREE_BUILD, where)
            b.upload_scene(scene)
            b.resize(96, 64)
            b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            b.set_counters(True)
            b.reset_stats()
            b.render(1, 2, 1)
            st = b.stats()
            img, _ = b.readback()
            seen.append((st["node_visits"], st["tri_tests"], st["max_stack"], img))
        finally:
            b.close()
    assert seen[0][:3] == seen[1][:3]
    assert seen[0][:2] != seen[2][:2]
    assert np.array_equal(seen[0][3], seen[1][3]) and np.array_equal(seen[0][3], seen[2][3])


def test_tree_options_may_be_flipped_on_a_committed_scene():
    """Every pair (who builds, which trees) on one context, one after the other: same image each time."""
    scene = capi.Scene("mixed", nu=96, nv=16)
    cam = capi.make_camera(width=64, height=48, bounces=4, ray_per_pixel=2)
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    b = capi.Backend(0)
    try:
        b.upload_scene(scene)
        b.resize(64, 48)
        for where in (capi.RT_TREE_BUILD_DEVICE, capi.RT_TREE_BUILD_HOST, capi.RT_TREE_BUILD_DEVICE):
            b.set_option(capi.RT_OPT_TREE_BUILD, where)
            for mode in (3, 0, 4, 2, 1, 3):
                b.set_option(capi.RT_OPT_WIDE_BVH, mode)
                assert b.info(capi.RT_INFO_TREE_MODE) == mode
                b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
                b.reset_accum()
                b.render(1, 2, 1)
                got, _ = b.readback()
                if mode in (0, 3, 4):   # (1 and 2 may let the other triangle of a distance tie win)
                    assert np.array_equal(got, want, equal_nan=True), f"where {where} mode {mode}"
    finally:
        b.close()


def _glb_of_triangles(path, positions):
    """One mesh of len(positions) / 3 triangles (positions (n, 3), three rows per triangle)."""
    from glb_writer import GlbBuilder, U32
    b = GlbBuilder()
    grey = b.material(pbrMetallicRoughness={"baseColorFactor": [0.7, 0.7, 0.7, 1], "metallicFactor": 0.0})
    positions = np.asarray(positions, np.float32)
    normals = np.tile(np.array([[0, 1, 0]], np.float32), (len(positions), 1))
    mesh = b.mesh([b.primitive(positions, np.arange(len(positions), dtype=np.uint32), grey, normals=normals, index_type=U32)])
    b.node(mesh=mesh, root=True)
    return b.write(path)


@pytest.mark.parametrize("kind", ["identical", "growing"])
def test_device_tree_rounds_stay_few_on_degenerate_models(tmp_path, kind):
    """120,000 copies of one triangle (every surface equal: the order among equals must pair them up at once) and
    120,000 triangles whose boxes grow steadily along the sorted order (every cluster prefers its smaller neighbour:
    the forced pairing after a round that merged next to nothing) -- either would take one round per triangle
    otherwise, seconds of launches.  Commit only: tracing 120,000 coincident triangles is no test of the build."""
    import time
    n = 120000
    if kind == "identical":
        tri = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32)
        positions = np.tile(tri, (n, 1))
    else:
        k = np.arange(n, dtype=np.float64)
        size = 1e-6 * 1.0001 ** k                 # box k reaches from the origin to size[k]: a chain of nested boxes
        positions = np.zeros((n, 3, 3), np.float64)
        positions[:, 1, 0] = size
        positions[:, 2, 2] = size
        positions[:, :, 1] = (k * 1e-9)[:, None]
        positions = positions.reshape(-1, 3)
    scene = capi.Scene("glb:" + str(_glb_of_triangles(tmp_path / f"{kind}.glb", positions)))
    b = capi.Backend(0)
    try:
        b.set_option(capi.RT_OPT_TREE_BUILD, capi.RT_TREE_BUILD_DEVICE)
        b.set_option(capi.RT_OPT_WIDE_BVH, 3)
        t0 = time.perf_counter()
        b.upload_scene(scene)
        took = time.perf_counter() - t0
        assert b.info(capi.RT_INFO_TREE_MODE) == 3
        assert took < 1.0, f"commit took {took:.2f} s"
    finally:
        b.close()


def test_device_tree_with_every_round_forced(monkeypatch):
    """The fallback pairing alone (neighbours of the sorted order merged whatever their surfaces) still gives a valid tree."""
    monkeypatch.setenv("RT_TREE_FORCE_PAIRS", "1")
    import subprocess, sys, os
    code = ("import numpy as np, sys; sys.path[:0] = ['.', 'tests'];"
            "from raytracer_amd import capi; from oracle_binding import OracleScene;"
            "s = capi.Scene('cornell_knot', nu=96, nv=16); c = capi.make_camera(width=48, height=36, bounces=3, ray_per_pixel=2);"
            "b = capi.Backend(0); b.set_option(capi.RT_OPT_TREE_BUILD, 1); b.upload_scene(s); b.resize(48, 36);"
            "b.set_frame(c, s.sky, capi.RT_TYPE_RGB); b.render(1, 2, 1); g, _ = b.readback(); b.close();"
            "w, _ = OracleScene(s).render(c, capi.RT_TYPE_RGB, 1, 2, 1); sys.exit(0 if np.array_equal(g, w, equal_nan=True) else 3)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ)).returncode == 0
