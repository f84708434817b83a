"""GPU parity: the HIP backend, called through the C ABI, against the CPU checker.

Bars (same seeded inputs on both sides):
  * primary-hit (model, identifier) ids: bit-exact, 100 % of pixels;
  * t, u, v: bit-exact (the stated requirement is 1e-5 relative);
  * radiance: per-pixel relative RMSE <= 1e-4 (the two sides share pinned math,
    so they are expected to be equal to the last bit; the tolerance is the
    stated bar, the stricter count of differing pixels is reported too).
"""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

SCENES = [
    ("cornell_quads", {}),
    ("roughness_demo", {}),
    ("three_color", {}),
    ("cornell_knot", dict(nu=128, nv=24)),
    ("knot_glass", dict(nu=128, nv=24)),
    ("prism", {}),
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),
    ("mixed", dict(nu=96, nv=16)),
]


@pytest.fixture(scope="module")
def backend():
    b = capi.Backend(0)
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
    """sqrt(mean((a-b)^2)) / max(mean(|b|), eps) over rgb."""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.mean(np.abs(b)), 1e-12))


@pytest.mark.parametrize("name,kw", SCENES)
def test_primary_hit_ids_bit_exact(backend, name, kw):
    W, H = 96, 72  # not a multiple of the 8x8 path tile in y? 72 = 9*8; 96 = 12*8
    scene, cam = _setup(backend, name, kw, W, H, 4, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = backend.trace_primary(frame=3)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=3)
    assert np.array_equal(gm, om), f"model ids differ at {np.argwhere(gm != om)[:5]}"
    assert np.array_equal(gi, oi), f"identifier ids differ at {np.argwhere(gi != oi)[:5]}"
    assert np.array_equal(gt, ot)
    assert np.array_equal(gu, ou)
    assert np.array_equal(gv, ov)


def test_primary_hit_ragged_resolution(backend):
    # Width/height that are not multiples of the 8x8 path tile.
    W, H = 70, 45
    scene, cam = _setup(backend, "mixed", dict(nu=64, nv=12), W, H, 2, 1, capi.RT_TYPE_RGB)
    gm, gi, gt, gu, gv = backend.trace_primary(frame=1)
    om, oi, ot, ou, ov, _ = OracleScene(scene).trace_primary(cam, frame=1)
    assert np.array_equal(gm, om) and np.array_equal(gi, oi) and np.array_equal(gt, ot)


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", SCENES)
def test_radiance_matches_checker(backend, name, kw, rtype):
    W, H, frames = 64, 48, 3
    scene, cam = _setup(backend, name, kw, W, H, 4, 4, rtype)
    backend.render(first_frame=1, n_frames=frames, stride=1)
    got, n = backend.readback()
    assert n == frames
    want, _ = OracleScene(scene).render(cam, rtype, 1, frames, 1)
    assert np.array_equal(got[..., 3], want[..., 3])
    differing = int(np.sum(np.any(got[..., :3] != want[..., :3], axis=-1)))
    err = rel_rmse(got, want)
    print(f"{name} type={rtype}: differing pixels {differing}/{W * H}, rel RMSE {err:.3e}")
    assert np.isfinite(got).all() == np.isfinite(want).all()
    assert err <= 1e-4
    assert differing == 0


def test_frame_stride_and_batching(backend):
    """Sample-split sharding: frames {2,5,8} must equal the checker's, and rendering them
    in one call or three must give the same sum (batches carry partial frames)."""
    W, H = 40, 40
    scene, cam = _setup(backend, "cornell_knot", dict(nu=64, nv=12), W, H, 3, 9, capi.RT_TYPE_RGB)
    backend.render(first_frame=2, n_frames=3, stride=3)
    one, _ = backend.readback()
    backend.reset_accum()
    for f in (2, 5, 8):
        backend.render(first_frame=f, n_frames=1, stride=1)
    three, n = backend.readback()
    assert n == 3
    want, _ = OracleScene(scene).render(cam, 0, 2, 3, 3)
    assert np.array_equal(one, want)
    assert np.array_equal(three, want)


def test_ray_counts_match_checker(backend):
    W, H = 48, 48
    scene, cam = _setup(backend, "mixed", dict(nu=64, nv=12), W, H, 4, 4, capi.RT_TYPE_RGB)
    backend.set_counters(True)
    backend.reset_stats()
    backend.render(1, 2, 1)
    st = backend.stats()
    backend.set_counters(False)
    _, ost = OracleScene(scene).render(cam, 0, 1, 2, 1)
    assert st["rays"] == ost["rays"]
    assert st["node_visits"] == ost["node_visits"]
    assert st["tri_tests"] == ost["tri_tests"]
    assert st["shaded_hits"] == ost["shaded_hits"]
    assert st["camera_samples"] == ost["camera_samples"]
