"""GPU: the colour tables and functions the kernels use (rt_eval_colour) against the CPU checker's,
value for value, and the closed-form white point on the device itself."""

import numpy as np
import pytest

from raytracer_amd import capi
import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    b = capi.Backend(0)
    yield b
    b.close()


def test_device_tables_equal_the_checkers(backend):
    assert np.array_equal(backend.eval_colour(4), ob.eval_colour(4))
    assert np.array_equal(backend.eval_colour(4, items=3)[2], ob.eval_colour(4)[0])


def test_device_colour_functions_equal_the_checkers(backend):
    rng = np.random.default_rng(7)
    lam = np.concatenate([rng.uniform(370, 790, 20000), 380 + 10 * np.arange(41), [379.999, 380.0, 569.99, 570, 579.99, 580, 769.99, 770, 779.99, 780, 780.01, np.nan]]).astype(np.float32)
    assert np.array_equal(backend.eval_colour(0, lam), ob.eval_colour(0, lam), equal_nan=True)
    basis = np.stack([np.tile(lam, 7), np.repeat(np.arange(7), lam.size)], 1).astype(np.float32)
    assert np.array_equal(backend.eval_colour(1, basis), ob.eval_colour(1, basis), equal_nan=True)
    rgb = rng.uniform(-0.2, 1.2, (lam.size, 3)).astype(np.float32)
    rgb[:200, 0] = rgb[:200, 1]              # ties between channels select other branches of the decomposition
    rgb[200:400, 1] = rgb[200:400, 2]
    rgb[400:500] = rgb[400:500, :1]
    refl = np.concatenate([lam[:, None], rgb], 1)
    assert np.array_equal(backend.eval_colour(2, refl), ob.eval_colour(2, refl), equal_nan=True)
    xyz = rng.uniform(-1, 2, (5000, 3)).astype(np.float32)
    assert np.array_equal(backend.eval_colour(3, xyz), ob.eval_colour(3, xyz))


def test_equal_energy_spectrum_is_white_on_the_device(backend):
    lam = np.linspace(380, 780, 40001, dtype=np.float32)
    xyz = backend.eval_colour(0, lam).astype(np.float64).mean(axis=0) * 400.0 / 106.856895
    assert np.allclose(backend.eval_colour(3, xyz[None, :].astype(np.float32))[0], 1.0, atol=3e-3)


def test_bad_arguments(backend):
    out = np.zeros(3, np.float32)
    assert capi.hip.rt_eval_colour(backend._ctx, 7, out.ctypes.data, 1, out.ctypes.data) == -1
    assert capi.hip.rt_eval_colour(backend._ctx, 0, None, 1, out.ctypes.data) == -1
    assert capi.hip.rt_eval_colour(backend._ctx, 0, out.ctypes.data, 0, out.ctypes.data) == -1
