#!/usr/bin/env python3
"""Differential sweep: many random scenes, HIP backend against the CPU checker, bit for bit.

    python tools/fuzz_sweep.py [first_seed] [count] [wide|wide2|wide3]
Prints one line per mismatch and a summary; exit status 1 if anything differed.  With a third
argument the backend walks its opt-in four-wide trees, where a differing pixel is legitimate only
as a distance tie between two triangles of one model; the sweep shows how often that happens.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from raytracer_amd import capi  # noqa: E402
from oracle_binding import OracleScene  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    be = capi.Backend(0)
    if len(sys.argv) > 3:   # "wide": four-wide trees from the uploaded ones, "wide2": the library's own, "wide3": those with exact ties
        be.set_option(capi.RT_OPT_WIDE_BVH, int(sys.argv[3][4:] or 1))
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        nu = int(rng.choice([8, 60, 500, 3000]))
        w, h = int(rng.integers(9, 70)), int(rng.integers(9, 50))
        bounces, rpp = int(rng.integers(0, 8)), int(rng.choice([1, 2, 4, 9]))
        scene = capi.Scene("fuzz", nu=nu, nv=int(rng.integers(2, 16)), seed=seed)
        pos = (float(rng.uniform(-0.9, 0.9)), float(rng.uniform(0.1, 1.9)), float(rng.uniform(-2.5, 0.9)))
        cam = capi.make_camera(position=pos, yaw=float(rng.uniform(0, 360)), pitch=float(rng.uniform(-60, 60)),
                               fov=float(rng.uniform(20, 120)), width=w, height=h, bounces=bounces, ray_per_pixel=rpp)
        be.upload_scene(scene)
        be.resize(w, h)
        orc = OracleScene(scene)
        for rtype in (capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL):
            # the reference's shader-compile-time switches, at random (one scene in four each)
            nee, strat, hero = bool(rng.integers(0, 4)), bool(rng.integers(0, 4)), not rng.integers(0, 4)
            be.set_option(capi.RT_OPT_NEE, int(nee))
            be.set_option(capi.RT_OPT_STRATIFIED, int(strat))
            be.set_option(capi.RT_OPT_HERO_WAVELENGTH, int(hero))
            be.set_frame(cam, scene.sky, rtype)
            be.reset_accum()
            frame = int(rng.integers(1, 1 << 20))
            be.render(first_frame=frame, n_frames=2, stride=3)
            got, _ = be.readback()
            want, _ = orc.render(cam, rtype, frame, 2, 3, nee=nee, stratified=strat, hero_wavelength=hero)
            if not np.array_equal(got, want, equal_nan=True):
                bad += 1
                diff = np.argwhere(np.any((got != want) & ~(np.isnan(got) & np.isnan(want)), axis=-1))
                print(f"MISMATCH seed={seed} type={rtype} nu={nu} {w}x{h} bounces={bounces} rpp={rpp} nee={nee} stratified={strat} hero={hero}: {len(diff)} pixels, first {diff[0]}", flush=True)
        scene.close()
    be.close()
    print(f"{count} scenes x 2 integrators: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
