"""Known-answer tests that pin the CPU checker (oracle/) without a GPU.

The reference ships no fixtures, so the checker is pinned by (a) an independent
pure-Python restatement of the integer RNG, (b) closed-form ray/sphere and
ray/quad intersections on the reference's own procedural scenes, (c) the camera
model evaluated in float64, and (d) libm for the pinned transcendental functions.
"""

import ctypes as C
import math

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene, lib

M32 = 0xFFFFFFFF


def py_pcg(v):
    # reference src/RayTracer.cpp:99-103
    state = (v * 747796405 + 2891336453) & M32
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & M32
    return ((word >> 22) ^ word) & M32


def py_hash_seed(x, y, frame, sample):
    # reference src/RayTracer.cpp:105-108 ('*' binds tighter than '^')
    h = ((x * 73856093) & M32) ^ ((y * 19349663) & M32) ^ ((frame * 83492791) & M32) ^ ((sample * 2654435761) & M32)
    return py_pcg(h)


def test_pcg_vectors():
    rng = np.random.default_rng(1)
    for v in [0, 1, 2, 0x7FFFFFFF, 0x80000000, M32] + [int(x) for x in rng.integers(0, 2**32, 200)]:
        assert lib.orc_pcg(v) == py_pcg(v)
    # literal known answers (computed from the formula above)
    assert py_pcg(0) == lib.orc_pcg(0)
    assert [py_pcg(i) for i in range(4)] == [lib.orc_pcg(i) for i in range(4)]


def test_hash_seed_vectors():
    rng = np.random.default_rng(2)
    for _ in range(200):
        x, y, f, s = (int(v) for v in rng.integers(0, 2**32, 4))
        assert lib.orc_hash_seed(x, y, f, s) == py_hash_seed(x, y, f, s)
    assert lib.orc_hash_seed(3, 5, 1, 0) == py_hash_seed(3, 5, 1, 0)


def test_rand_float_range_and_rounding():
    seed = C.c_uint32(12345)
    s = 12345
    for _ in range(100):
        got = lib.orc_rand_float(C.byref(seed))
        s = py_pcg(s)
        assert seed.value == s
        want = np.float32(np.float32(s) / np.float32(4294967296.0))
        assert np.float32(got) == want
        assert 0.0 <= got <= 1.0
    # float(u32) rounds to nearest: 0xFFFFFFFF -> 2^32 -> exactly 1.0 (documented closed interval)
    assert np.float32(0xFFFFFFFF) / np.float32(4294967296.0) == np.float32(1.0)


@pytest.mark.parametrize("fn,ref,lo,hi", [
    (0, math.sin, 0.0, 6.2832), (1, math.cos, 0.0, 6.2832), (2, math.tan, 0.05, 1.3),
    (3, math.exp, -20.0, 5.0), (4, math.log, 1e-6, 50.0), (8, math.sqrt, 0.0, 100.0),
])
def test_pinned_math_close_to_libm(fn, ref, lo, hi):
    xs = np.linspace(lo, hi, 4001, dtype=np.float32)
    got = np.array([lib.orc_math(fn, float(x), 0.0) for x in xs], dtype=np.float64)
    want = np.array([ref(float(x)) for x in xs], dtype=np.float64)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert err.max() < 4e-7, (fn, err.max())


def test_pinned_pow_and_minmax_nan():
    xs = np.linspace(0.0, 1.0, 1001, dtype=np.float32)
    got = np.array([lib.orc_math(5, float(x), 5.0) for x in xs])
    assert np.allclose(got, xs.astype(np.float64) ** 5, rtol=2e-6, atol=1e-30)
    assert lib.orc_math(5, 0.0, 5.0) == 0.0
    nan = float("nan")
    # IEEE minNum/maxNum: a NaN operand yields the other operand
    assert lib.orc_math(6, nan, 2.0) == 2.0 and lib.orc_math(6, 2.0, nan) == 2.0
    assert lib.orc_math(7, nan, -3.0) == -3.0 and lib.orc_math(7, -3.0, nan) == -3.0
    assert lib.orc_math(6, 1.0, 2.0) == 1.0 and lib.orc_math(7, 1.0, 2.0) == 2.0


def test_camera_ray_matches_float64_model():
    """Fragment main() of the reference (src/RayTracer.cpp:543-578) evaluated in float64."""
    W, H, frame = 160, 100, 7
    cam = capi.make_camera(width=W, height=H, bounces=2, ray_per_pixel=4)
    pos = np.array(list(cam.position), dtype=np.float64)
    fwd, right, up = (np.array(list(v), dtype=np.float64) for v in (cam.forward, cam.right, cam.up))
    for (px, py) in [(0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (W // 2, H // 2), (17, 63)]:
        out = (C.c_float * 6)()
        lib.orc_camera_ray(C.byref(cam), px, py, frame, out)
        seed = py_hash_seed(px, py, frame, 0)
        seed = py_pcg(seed); jx = float(np.float32(seed)) / 4294967296.0
        seed = py_pcg(seed); jy = float(np.float32(seed)) / 4294967296.0
        focal = np.linalg.norm(fwd)
        vh = 2.0 * math.tan((3.1415926 / 180.0) * cam.fov * 0.5) * focal
        vw = vh * (W / H)
        ux, uy = px / W * 2 - 1, py / H * 2 - 1
        ssq = 2
        p = vw * 0.5 * ux * right + vh * 0.5 * uy * up + focal * fwd + pos
        d = p + ((0 + jx) / ssq) / W * right + ((0 + jy) / ssq) / H * up - pos
        d /= np.linalg.norm(d)
        assert np.allclose(list(out)[:3], pos, atol=0)
        assert np.allclose(list(out)[3:], d, atol=3e-6), (px, py)


def test_camera_basis_of_default_camera():
    # yaw 90, pitch 0 looks down +z with +x to the right and +y up (src/RayCamera.h:20-28)
    cam = capi.make_camera()
    assert np.allclose(list(cam.forward), [0, 0, 1], atol=1e-6)
    assert np.allclose(list(cam.right), [1, 0, 0], atol=1e-6)
    assert np.allclose(list(cam.up), [0, 1, 0], atol=1e-6)


def _sphere_t(o, d, c, r):
    oc = np.array(c) - np.array(o)
    a = np.dot(d, d); h = np.dot(d, oc); cc = np.dot(oc, oc) - r * r
    disc = h * h - a * cc
    if disc < 0:
        return None
    for t in ((h - math.sqrt(disc)) / a, (h + math.sqrt(disc)) / a):
        if t > 1e-8:
            return t
    return None


def test_closed_form_sphere_and_quad_hits():
    """ThreeColorDemo (src/RaySceneBuilder.cpp:35-70): sphere (0,0,1) r=.12, floor quad y=-0.1."""
    scene = capi.Scene("three_color")
    assert (scene.object_count, scene.models_count, scene.light_sources_count) == (5, 0, 0)
    orc = OracleScene(scene)
    # straight at the white sphere
    h = orc.hit((0.0, 0.0, -1.0), (0.0, 0.0, 1.0))
    assert h["model"] == -2 and h["ident"] == 0
    assert abs(h["t"] - 1.88) < 1e-6
    assert np.allclose(h["normal"], (0, 0, -1), atol=1e-6) and h["front_face"]
    # oblique ray at the sphere: closed form
    o, d = (0.05, 0.3, -0.5), np.array([-0.02, -0.18, 1.0]); d = d / np.linalg.norm(d)
    h = orc.hit(o, tuple(d))
    t = _sphere_t(o, d, (0, 0, 1), 0.12)
    assert t is not None and h["ident"] == 0 and abs(h["t"] - t) < 2e-6
    # down onto the floor quad (primitive 1): t = (0.5 - (-0.1)) / 1
    h = orc.hit((2.0, 0.5, 3.0), (0.0, -1.0, 0.0))
    assert h["model"] == -2 and h["ident"] == 1 and abs(h["t"] - 0.6) < 1e-6
    assert np.allclose(h["normal"], (0, 1, 0), atol=1e-6)
    # from below the quad the normal flips toward the ray
    h = orc.hit((2.0, -1.0, 3.0), (0.0, 1.0, 0.0))
    assert h["ident"] == 1 and abs(h["t"] - 0.9) < 1e-6 and np.allclose(h["normal"], (0, -1, 0), atol=1e-6)
    # outside the quad's extent, pointing away from everything: miss
    h = orc.hit((50.0, 5.0, 50.0), (0.0, 1.0, 0.0))
    assert h["t"] >= 1e20 and h["model"] == -1
    # grazing: starting inside the sphere uses the far root
    h = orc.hit((0.0, 0.0, 1.0), (0.0, 0.0, 1.0))
    assert h["ident"] == 0 and abs(h["t"] - 0.12) < 1e-6 and not h["front_face"]


def test_cull_face_quad():
    """cornell_quads has a culled front quad (primitive 6): invisible from the camera side."""
    scene = capi.Scene("cornell_quads")
    orc = OracleScene(scene)
    h = orc.hit((0.0, 1.0, -2.2), (0.0, 0.0, 1.0))
    assert h["ident"] != 6          # passes through the culled front face
    h = orc.hit((0.9, 1.9, 0.9), (0.0, 0.0, -1.0))
    assert h["ident"] == 6 and abs(h["t"] - 1.9) < 1e-5   # hit from inside


def test_model_triangle_hit_and_tie_rule():
    """Cornell box as triangles: the back wall (z=+1) is hit at t=3.2 from the default camera;
    a ray along the shared diagonal of a wall's two triangles reports the first triangle in leaf order."""
    scene = capi.Scene("cornell_knot", nu=8, nv=4)
    orc = OracleScene(scene)
    h = orc.hit((0.9, 1.9, -2.2), (0.0, 0.0, 1.0))
    assert h["model"] == 0 and abs(h["t"] - 3.2) < 1e-5
    assert np.allclose(np.abs(h["normal"]), (0, 0, 1), atol=1e-6)


def test_oracle_mean_matches_sum_for_reference_progression():
    """frameCount = 1..n running mean (src/RayTracer.cpp:596) equals sum/n up to fp32 rounding."""
    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    cam = capi.make_camera(width=24, height=16, bounces=3, ray_per_pixel=1)
    rgba, mean, _ = OracleScene(scene).render(cam, 0, 1, 5, 1, want_mean=True)
    assert np.allclose(mean[..., :3], rgba[..., :3] / 5.0, rtol=2e-6, atol=1e-7)
    assert np.all(rgba[..., 3] == 5.0)


@pytest.mark.parametrize("name,kw", [("cornell_knot", dict(nu=64, nv=12)), ("unlit_knot", dict(nu=96, nv=10)), ("fuzz", dict(nu=300, nv=6, seed=5))])
def test_tree_traversal_finds_what_brute_force_finds(name, kw):
    """Independent check of the host tree build + the checker's traversal and triangle test: the
    closest hit of every primary ray, by float64 Moller-Trumbore over ALL triangles decoded from
    the serialised streams (no tree), must be the checker's hit.  Pixels whose best and second
    best distances are within 1e-6 of each other are excluded (ties and silhouettes)."""
    from test_host_layer import decode_model, i32
    scene = capi.Scene(name, **kw)
    W, H = 48, 36
    cam = capi.make_camera(width=W, height=H, bounces=1, ray_per_pixel=1)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=1)

    tris, owner, alpha = [], [], []
    info = i32(scene.streams[3]).reshape(-1, 4)
    for m in range(scene.models_count):
        d = decode_model(scene, m)
        end = info[m + 1][3] if m + 1 < scene.models_count else scene.streams[4].size
        verts = scene.streams[4][d["vert_at"]: end].reshape(-1, 8)[:, :3].astype(np.float64)
        mats = scene.streams[2].reshape(-1, 23)
        for k, ident in enumerate(d["ident"]):
            tris.append(verts[ident[:3]])
            owner.append((m, k))
            mat = mats[ident[3]]
            alpha.append(bool(d["ident_f"][k, 4] != 0) and i32(mat[:6])[1] != -1 and mat[20] > 0)
    tris = np.array(tris)                       # (T, 3, 3)
    alpha = np.array(alpha)                     # triangles that a texture lookup may make transparent
    A, ab, ac = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    nrm = np.cross(ab, ac)

    checked = 0
    for y in range(H):
        for x in range(W):
            out = (C.c_float * 6)()
            lib.orc_camera_ray(C.byref(cam), x, y, 1, out)
            ro, rd = np.array(out[:3], np.float64), np.array(out[3:], np.float64)
            det = -(nrm @ rd)
            ok = np.abs(det) > 1e-12
            ao = ro - A
            dao = np.cross(ao, rd)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.einsum("ij,ij->i", ao, nrm) / det
                u = np.einsum("ij,ij->i", ac, dao) / det
                v = -np.einsum("ij,ij->i", ab, dao) / det
                hit = ok & (t > 0) & (u >= 0) & (v >= 0) & (u + v <= 1)
            ts = np.where(hit & ~alpha, t, np.inf)
            if (hit & alpha & (t <= ts.min() + 1e-6)).any():
                continue                                   # an alpha-cut triangle in front: its fate depends on a texel
            order = np.argsort(ts)
            best, second = ts[order[0]], ts[order[1]]
            margin_uv = min(u[order[0]], v[order[0]], 1 - u[order[0]] - v[order[0]]) if np.isfinite(best) else 1.0
            if scene.object_count and om[y, x] == -2:
                continue                                   # an analytic primitive is closer; covered by the closed-form tests
            if not np.isfinite(best):
                if scene.object_count == 0:
                    assert om[y, x] == -1, (x, y)
                continue
            if second - best < 1e-6 or margin_uv < 1e-5:
                continue
            if scene.object_count and ot[y, x] < best - 1e-5:
                continue                                   # a primitive in front (reported as model -2 above) or beside
            checked += 1
            assert (om[y, x], oi[y, x]) == owner[order[0]], (x, y, best)
            assert abs(ot[y, x] - best) <= 1e-5 * best
            assert abs(ou[y, x] - u[order[0]]) <= 2e-4 and abs(ov[y, x] - v[order[0]]) <= 2e-4
    assert checked > 100                       # enough pixels really went through the comparison


def test_white_furnace():
    """A white sphere in a uniformly white environment (sanity of pdfs and weights, not a pin):
    clear glass returns the environment exactly in RGB, a matte white sphere within the few
    per cent the Disney diffuse term adds at grazing angles, rough white metal loses a little."""
    cam = capi.make_camera(width=48, height=36, bounces=12, ray_per_pixel=16)
    means = {}
    for kind in (0, 1, 2):
        orc = OracleScene(capi.Scene("furnace", nu=kind))
        on = orc.trace_primary(cam)[0] == -2
        img, _ = orc.render(cam, 0, 1, 4, 1)
        means[kind] = float((img[..., :3] / 4)[on].mean())
    assert means[2] == pytest.approx(1.0, abs=1e-6)
    assert 0.98 < means[0] < 1.12
    assert 0.90 < means[1] < 1.02


def test_hero_wavelength_mode_estimates_the_same_image():
    """HERO_WAVELENGTH_ENABLE 1 (src/RayTracer.cpp:1357-1364) draws four wavelengths per camera
    sample instead of one: a different estimator of the same integral.  On a matte scene lit by the
    sky the two means agree within the noise, and with one wavelength per sample the hero mode needs
    about four times the rays."""
    scene = capi.Scene("furnace", nu=0)
    orc = OracleScene(scene)
    cam = capi.make_camera(width=24, height=18, bounces=3, ray_per_pixel=4)
    plain, st0 = orc.render(cam, 1, 1, 32, 1)
    hero, st1 = orc.render(cam, 1, 1, 8, 1, hero_wavelength=True)
    assert st1["camera_samples"] * 4 == st0["camera_samples"]
    assert 0.9 < st1["rays"] / st0["rays"] < 1.1           # 8 frames x 4 wavelengths against 32 frames
    a, b = plain[..., :3] / 32.0, hero[..., :3] / 8.0
    assert np.isfinite(b).all()
    assert abs(a.mean() - b.mean()) < 0.03 * a.mean()
    # and it is a different random sequence, not the same image
    assert not np.array_equal(plain / 32.0, hero / 8.0)


def test_hero_wavelengths_cover_the_range_in_quarters():
    """get_hero_wavelengths (src/RayTracer.cpp:1259-1272): four wavelengths a quarter of the range
    apart, wrapped into [380, 780).  Seen from outside through an emissive white quad filling the
    view: the pixel is the mean of four CIE samples, so over many frames it tends to the white of
    the plain mode much faster than one wavelength per sample does."""
    scene = capi.Scene("furnace", nu=2)
    orc = OracleScene(scene)
    cam = capi.make_camera(width=8, height=6, bounces=1, ray_per_pixel=1)
    plain, _ = orc.render(cam, 1, 1, 16, 1)
    hero, _ = orc.render(cam, 1, 1, 16, 1, hero_wavelength=True)
    sp, sh = (plain[..., :3] / 16).std(axis=(0, 1)).mean(), (hero[..., :3] / 16).std(axis=(0, 1)).mean()
    assert sh < 0.7 * sp                                   # stratified over the spectrum: less colour noise


def test_without_nee_the_estimate_is_the_same_but_noisier():
    """ENABLE_NEE 0 (src/RayTracer.cpp:24): paths find the light only by running into it.  Same
    expectation as with NEE + MIS, much more variance; no shadow rays."""
    scene = capi.Scene("cornell_knot", nu=32, nv=8)      # the light is a model: only those are light-sampled
    orc = OracleScene(scene)
    cam = capi.make_camera(width=24, height=18, bounces=4, ray_per_pixel=4)
    a, st_a = orc.render(cam, 0, 1, 48, 1)
    b, st_b = orc.render(cam, 0, 1, 48, 1, nee=False)
    assert st_b["rays"] < st_a["rays"]
    ma, mb = a[..., :3].mean() / 48, b[..., :3].mean() / 48
    assert abs(ma - mb) < 0.15 * ma
    assert (b[..., :3] / 48).std() > (a[..., :3] / 48).std()
