"""GPU parity on the corners of the input space: empty and unlit scenes, zero bounces, one-pixel
and ragged images, non-square sample counts, cameras inside geometry or looking away from it."""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[3, 0], ids=["default_trees", "reference_order_walk"])
def backend(request):
    """Every test runs twice: on the library's default (its own four-wide trees, rays with a distance
    tie traced again in the reference's order) and on the plain reference-order walk of the uploaded
    trees (RT_OPT_WIDE_BVH 0).  Both have the same bar: the checker's result."""
    b = capi.Backend(0)
    if request.param != 3:
        b.set_option(capi.RT_OPT_WIDE_BVH, request.param)
    b.tree_mode = request.param
    yield b
    b.close()


def both(backend, name, kw, cam_kw, rtype, frames=2, first=1):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(**cam_kw)
    backend.upload_scene(scene)
    backend.resize(cam_kw["width"], cam_kw["height"])
    backend.set_frame(cam, scene.sky, rtype)
    ids_gpu = backend.trace_primary(frame=first)
    backend.render(first_frame=first, n_frames=frames, stride=1)
    got, n = backend.readback()
    orc = OracleScene(scene)
    ids_cpu = orc.trace_primary(cam, frame=first)
    want, st = orc.render(cam, rtype, first, frames, 1)
    assert n == frames
    for g, c in zip(ids_gpu, ids_cpu):
        assert np.array_equal(g, c)
    assert np.array_equal(got, want, equal_nan=True)
    return scene, got, ids_gpu


TYPES = [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL]


@pytest.mark.parametrize("rtype", TYPES)
def test_empty_scene_is_sky_only(backend, rtype):
    scene, img, ids = both(backend, "empty", {}, dict(width=33, height=17, bounces=4, ray_per_pixel=1), rtype)
    assert scene.object_count == 0 and scene.models_count == 0
    assert (ids[0] == -1).all() and np.isfinite(img).all()
    if rtype == capi.RT_TYPE_RGB:
        assert (img[..., :3] > 0).all()       # a single-wavelength Spectral sample may have negative RGB components


@pytest.mark.parametrize("rtype", TYPES)
def test_model_without_any_light_source(backend, rtype):
    scene, img, ids = both(backend, "unlit_knot", dict(nu=64, nv=12), dict(width=40, height=30, bounces=3, ray_per_pixel=4), rtype)
    assert scene.light_sources_count == 0
    assert (ids[0] >= 0).any() and np.isfinite(img).all()


@pytest.mark.parametrize("rtype", TYPES)
@pytest.mark.parametrize("bounces", [0, 1, 16])
def test_bounce_limits(backend, rtype, bounces):
    both(backend, "mixed", dict(nu=32, nv=8), dict(width=32, height=24, bounces=bounces, ray_per_pixel=1), rtype)


@pytest.mark.parametrize("w,h", [(1, 1), (1, 9), (9, 1), (7, 3), (8, 8), (65, 9)])
def test_tiny_and_ragged_images(backend, w, h):
    both(backend, "cornell_knot", dict(nu=32, nv=8), dict(width=w, height=h, bounces=2, ray_per_pixel=1), capi.RT_TYPE_RGB)


@pytest.mark.parametrize("rtype", TYPES)
@pytest.mark.parametrize("rpp", [2, 5, 16, 26])
def test_sample_counts_that_are_not_squares(backend, rtype, rpp):
    """ssq = int(sqrt(rayPerPixel)): 2 -> 1x1, 5 -> 2x2, 26 -> 5x5; the Spectral wavelength
    stratum width still divides by rayPerPixel itself."""
    both(backend, "cornell_quads", {}, dict(width=24, height=16, bounces=2, ray_per_pixel=rpp), rtype, frames=1)


CAMERAS = {
    "inside_the_knot_tube": dict(position=(0.5, 0.95, 0.1)),
    "outside_looking_away": dict(position=(0.0, 1.0, -2.2), yaw=-90.0),
    "looking_straight_up": dict(position=(0.0, 0.5, 0.0), pitch=89.0),
    "grazing_the_floor": dict(position=(0.0, 1e-3, -0.9), pitch=0.0),
    "wide_fov": dict(fov=150.0),
    "narrow_fov": dict(fov=2.0),
    "behind_the_back_wall": dict(position=(0.0, 1.0, 1.5)),
}


@pytest.mark.parametrize("case", sorted(CAMERAS))
def test_camera_placements(backend, case):
    cam = dict(width=48, height=32, bounces=3, ray_per_pixel=1)
    cam.update(CAMERAS[case])
    both(backend, "cornell_knot", dict(nu=96, nv=16), cam, capi.RT_TYPE_RGB)


def test_frame_numbers_near_the_u32_limit(backend):
    both(backend, "three_color", {}, dict(width=16, height=16, bounces=2, ray_per_pixel=1), capi.RT_TYPE_RGB, frames=2, first=0xFFFFFFF0)


def test_bad_arguments_are_refused(backend):
    scene = capi.Scene("three_color")
    backend.upload_scene(scene)
    backend.resize(16, 16)
    for kw in (dict(ray_per_pixel=0), dict(bounces=-1), dict(bounces=10 ** 6), dict(width=17)):
        args = dict(width=16, height=16, bounces=2, ray_per_pixel=1)
        args.update(kw)
        with pytest.raises(capi.RtError):
            backend.set_frame(capi.make_camera(**args), scene.sky, capi.RT_TYPE_RGB)
    with pytest.raises(capi.RtError):
        backend.resize(0, 16)
    backend.set_frame(capi.make_camera(width=16, height=16), scene.sky, capi.RT_TYPE_RGB)
    with pytest.raises(capi.RtError):
        backend.render(first_frame=1, n_frames=1, stride=0)


def test_many_renders_without_a_sync_keep_the_timing_bookkeeping_bounded(backend):
    """3000 one-frame render calls back to back (an interactive loop that never reads back):
    the sum is still exact and the device time is still accounted for."""
    scene = capi.Scene("three_color")
    cam = capi.make_camera(width=16, height=16, bounces=1, ray_per_pixel=1)
    backend.upload_scene(scene)
    backend.resize(16, 16)
    backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    backend.reset_stats()
    for f in range(3000):
        backend.render(first_frame=1 + f, n_frames=1, stride=1)
    got, n = backend.readback()
    st = backend.stats()
    assert n == 3000 and st["trace_launches"] == 3000 * 3 and st["total_ms"] > 0
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 3000, 1)
    assert np.array_equal(got, want)


def test_inconsistent_streams_are_refused_and_leave_no_scene(backend):
    """Counts that do not match the uploaded streams, child references that point backwards and
    vertex indices beyond the stream are caught at commit; rendering afterwards is a state error."""
    import ctypes as C
    hip = capi.hip
    scene = capi.Scene("cornell_knot", nu=16, nv=6)
    ctx = backend._ctx

    def upload(streams):
        for slot in range(1, 6):
            a = np.ascontiguousarray(streams[slot], np.float32)
            assert hip.rt_upload(ctx, slot, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0

    good = {s: scene.streams[s].copy() for s in range(1, 6)}
    upload(good)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count + 1, scene.light_sources_count) == -5   # more models than info records
    assert hip.rt_render(ctx, 1, 1, 1) == -2

    info = good[3].view(np.int32).reshape(-1, 4)
    bad = {s: v.copy() for s, v in good.items()}
    node0 = int(info[1][3]) + int(info[1][1]) * 5                      # word offset of the knot's root node
    bad[4].view(np.int32)[node0 + 6] = 0                               # left child = itself: a cycle
    upload(bad)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == -5

    bad = {s: v.copy() for s, v in good.items()}
    ident0 = node0 + int(info[1][2]) * 9
    bad[4].view(np.int32)[ident0] = 1 << 28                            # vertex index far beyond the stream
    upload(bad)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == -5

    bad = {s: v.copy() for s, v in good.items()}
    bad[2] = bad[2][:23]                                               # identifiers name materials that are gone
    upload(bad)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == -5

    # counts whose section offsets wrap in 32-bit arithmetic (858993460 * 5 = 4 (mod 2^32)) must not
    # pass as small offsets
    for column in (0, 1, 2):
        bad = {s: v.copy() for s, v in good.items()}
        bad[3].view(np.int32).reshape(-1, 4)[1][column] = 858993460 if column != 2 else 477218589   # * 9 wraps too
        upload(bad)
        assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == -5

    # two parents sharing one child: a graph, which would be walked as an exponentially large tree
    bad = {s: v.copy() for s, v in good.items()}
    words = bad[4].view(np.int32)
    root_left, root_right = int(words[node0 + 6]), int(words[node0 + 7])
    assert bad[4][node0 + root_left * 9 + 8] == 0.0                    # the root's left child is an inner node
    words[node0 + root_left * 9 + 7] = root_right                      # ... which now also claims the root's right child
    upload(bad)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == -5

    upload(good)
    assert hip.rt_commit_scene(ctx, scene.object_count, scene.models_count, scene.light_sources_count) == 0


@pytest.mark.parametrize("rtype", TYPES)
@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_soups_with_random_materials(backend, rtype, seed):
    """Differential test: random triangles (some degenerate), skewed normals, every material field
    at random including the corner values, random textures and analytic primitives."""
    nu = 40 if seed % 4 == 0 else 400          # shallow and deeper trees
    _, img, _ = both(backend, "fuzz", dict(nu=nu, nv=10, seed=seed), dict(width=56, height=40, bounces=6, ray_per_pixel=4), rtype)


def test_host_layer_hands_backend_errors_on_instead_of_trapping():
    """A camera the backend refuses (more bounces than it keeps stages for) through the C++ host
    layer: RayEngine::render records the status, rth_engine_render returns it, the process lives."""
    import ctypes as C
    pos = (C.c_float * 3)(0.0, 1.0, -2.2)
    eng = capi.host.rth_engine_create(b"cornell_quads", None, 0, pos, 90.0, 0.0, 50.0, 32, 24, 100, 1, capi.RT_TYPE_RGB, 0)
    assert eng
    try:
        assert capi.host.rth_engine_render(eng, 1) == -1          # RT_ERR_INVALID from rt_set_frame
        out = np.empty((24, 32, 4), np.float32)
        assert capi.host.rth_engine_frame(eng, out.ctypes.data_as(C.c_void_p), out.size) == 0   # nothing was rendered
        assert not out.any()
    finally:
        capi.host.rth_engine_free(eng)
    assert not capi.host.rth_engine_create(b"cornell_quads", None, 0, pos, 90.0, 0.0, 50.0, 0, 24, 3, 1, capi.RT_TYPE_RGB, 0)


def test_reserve_sizes_the_path_buffers_ahead_of_the_render(backend):
    """rt_reserve before rt_set_frame is a state error; after it, the render call of that size finds
    its buffers in place (same image as without the reservation)."""
    W, H = 40, 32
    b = capi.Backend(0)
    try:
        with pytest.raises(capi.RtError):
            b.reserve(2)
        scene = capi.Scene("cornell_knot", nu=32, nv=8)
        cam = capi.make_camera(width=W, height=H, bounces=3, ray_per_pixel=4)
        b.upload_scene(scene)
        b.resize(W, H)
        b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        with pytest.raises(capi.RtError):
            b.reserve(0)
        b.reserve(3)
        b.render(1, 3, 1)
        got, n = b.readback()
    finally:
        b.close()
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 3, 1)
    assert n == 3 and np.array_equal(got, want)
