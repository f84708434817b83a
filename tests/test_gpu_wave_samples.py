"""GPU: the order of the paths inside a batch of several sample planes.

A wave holds B samples of 64 / B pixels, B = the largest power of two up to 16 (or the RT_WAVE_SAMPLES value) that
divides the batch's planes: 16 samples of a 2x2-pixel block for 16, 32, 48 ... planes, 4 samples of a 4x4 block for
12, one sample of the 8x8 tile for an odd count (Batch::waveSamples, pathIndex / pathOf in csrc/rt_wavefront.h).  Which slot a
(pixel, sample) pair occupies decides nothing about its value: seeds come from pixel, frame and sample
(src/RayTracer.cpp:572), and the frame sum adds samples in sample order.  Bar: the checker's image bit for bit, for
both integrators, hero wavelengths, ragged image sizes, tile-split shards and batches cut by the path ceiling.
"""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


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


# (samples per frame, frames): 16, 16, 32, 64 and 48 planes in one batch (16 per wave), 12 (4), 24 (8), 18 (2), 17 (1)
CASES = [(4, 4), (16, 1), (4, 8), (16, 4), (16, 3), (4, 3), (4, 6), (9, 2), (1, 17)]


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("rpp,frames", CASES)
def test_blocks_of_samples_per_wave_match_the_checker(backend, rpp, frames, rtype):
    scene, cam = _setup(backend, "mixed", dict(nu=48, nv=8), 50, 37, 3, rpp, rtype)   # ragged: 7 x 5 tiles, partly empty
    backend.reset_accum()
    backend.reset_stats()
    backend.render(2, frames, 3)
    got, n = backend.readback()
    want, ost = OracleScene(scene).render(cam, rtype, 2, frames, 3)
    assert n == frames and backend.stats()["rays"] == ost["rays"]
    assert np.array_equal(got, want, equal_nan=True), f"{int(np.any(got != want, axis=-1).sum())} differing pixels"


def test_hero_wavelengths_with_blocks_of_samples(backend):
    scene, cam = _setup(backend, "knot_glass", dict(nu=64, nv=12), 40, 24, 3, 4, capi.RT_TYPE_SPECTRAL)
    backend.set_option(capi.RT_OPT_HERO_WAVELENGTH, 1)
    try:
        backend.set_frame(cam, scene.sky, capi.RT_TYPE_SPECTRAL)
        backend.reset_accum()
        backend.render(1, 4, 1)
        got, _ = backend.readback()
    finally:
        backend.set_option(capi.RT_OPT_HERO_WAVELENGTH, 0)
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_SPECTRAL, 1, 4, 1, hero_wavelength=True)
    assert np.array_equal(got, want, equal_nan=True)


def test_unstratified_branch_with_blocks_of_samples(backend):
    scene, cam = _setup(backend, "cornell_knot", dict(nu=64, nv=12), 48, 32, 3, 16, capi.RT_TYPE_RGB)
    backend.set_option(capi.RT_OPT_STRATIFIED, 0)
    try:
        backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        backend.reset_accum()
        backend.render(1, 2, 1)
        got, _ = backend.readback()
    finally:
        backend.set_option(capi.RT_OPT_STRATIFIED, 1)
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1, stratified=False)
    assert np.array_equal(got, want, equal_nan=True)


def test_tile_shards_and_cut_batches_with_blocks_of_samples(backend):
    """32 planes: whole, as three tile-split shards, and with the path ceiling cutting the batch into pieces of 16,
    of 5 (then 4 + 1 samples per wave) and of 1 planes: the same image every time."""
    W, H = 70, 45
    scene, cam = _setup(backend, "mixed", dict(nu=64, nv=12), W, H, 3, 4, capi.RT_TYPE_RGB)
    backend.reset_accum()
    backend.render(1, 8, 1)
    full, _ = backend.readback()
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 8, 1)
    assert np.array_equal(full, want)
    total = np.zeros_like(full)
    for k in range(3):
        backend.set_tile_shard(k, 3)
        backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        backend.reset_accum()
        backend.render(1, 8, 1)
        total += backend.readback()[0]
    backend.set_tile_shard(0, 1)
    backend.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    assert np.array_equal(total, full)
    npt = ((W + 7) // 8) * ((H + 7) // 8) * 64
    try:
        for planes in (16, 5, 1):
            backend.set_option(capi.RT_OPT_MAX_PATHS, planes * npt)
            backend.reset_accum()
            backend.render(1, 8, 1)
            assert np.array_equal(backend.readback()[0], full), f"batches of {planes} planes"
    finally:
        backend.set_option(capi.RT_OPT_MAX_PATHS, 1 << 27)


def test_samples_per_wave_setting_does_not_change_the_image(monkeypatch):
    """RT_WAVE_SAMPLES 1 (one sample of 64 pixels per wave), 4, 64: same image as the default."""
    scene = capi.Scene("cornell_knot", nu=96, nv=16)
    cam = capi.make_camera(width=64, height=48, bounces=4, ray_per_pixel=16)
    images = []
    for value in (None, "1", "4", "64"):
        if value is None:
            monkeypatch.delenv("RT_WAVE_SAMPLES", raising=False)
        else:
            monkeypatch.setenv("RT_WAVE_SAMPLES", value)
        b = capi.Backend(0)
        try:
            b.upload_scene(scene)
            b.resize(64, 48)
            b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
            b.render(1, 4, 1)
            images.append(b.readback()[0])
        finally:
            b.close()
    for img in images[1:]:
        assert np.array_equal(img, images[0])
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 4, 1)
    assert np.array_equal(images[0], want)
