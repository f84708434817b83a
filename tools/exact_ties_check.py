#!/usr/bin/env python3
"""RT_OPT_WIDE_BVH 3 (the library's four-wide trees, rays with a distance tie traced again in the
reference's order) against the default reference-order walk, both on the GPU, at full image sizes:
every pixel of every image is expected to be equal bit for bit.  Prints the number of differing
pixels and of re-traced rays per case; exit status 1 if any pixel differs.

    python tools/exact_ties_check.py [spp]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CASES = [
    ("cornell_knot", dict(nu=2048, nv=213), 1920, 1080, capi.RT_TYPE_RGB),
    ("cornell_knot", dict(nu=2048, nv=213), 1920, 1080, capi.RT_TYPE_SPECTRAL),
    ("knot_glass", dict(nu=2048, nv=213), 1920, 1080, capi.RT_TYPE_SPECTRAL),
    ("interior", dict(nu=512, nv=64, tex_size=256), 1920, 1080, capi.RT_TYPE_RGB),
    ("textured_room", dict(nu=256, nv=32, tex_size=128), 1280, 720, capi.RT_TYPE_RGB),
    ("mixed", dict(nu=512, nv=64), 1280, 720, capi.RT_TYPE_RGB),
    ("ties", dict(nu=64, nv=5, seed=3), 1280, 720, capi.RT_TYPE_RGB),
    ("fuzz", dict(nu=20000, nv=9, seed=5), 1280, 720, capi.RT_TYPE_RGB),
]

bad = 0
for name, kw, W, H, rtype in CASES:
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=SPP)
    out = {}
    for mode in (0, 3):
        be = capi.Backend(0)
        be.set_option(capi.RT_OPT_WIDE_BVH, mode)
        be.upload_scene(scene)
        be.resize(W, H)
        be.set_frame(cam, scene.sky, rtype)
        be.reset_stats()
        t0 = time.perf_counter()
        be.render(1, 1, 1)
        img, _ = be.readback()
        out[mode] = (img, be.stats(), time.perf_counter() - t0)
        be.close()
    (img0, st0, dt0), (img3, st3, dt3) = out[0], out[3]
    differing = int(np.any(img0.view(np.uint32) != img3.view(np.uint32), axis=-1).sum())
    print(f"{name:14s} {W}x{H} type={rtype} {SPP} spp: {differing} differing pixels of {W * H}; "
          f"{st3['retraced']} of {st3['rays']} rays traced again; trace {st0['trace_ms']:.0f} -> {st3['trace_ms']:.0f} ms", flush=True)
    bad += differing != 0
sys.exit(1 if bad else 0)
