#!/usr/bin/env python3
"""Four-wide tree against the reference's traversal order on the same scenes: primary hits,
images and visit counts.  Differences can only be distance ties between triangles."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

CASES = [
    ("cornell_knot", dict(nu=256, nv=32), 320, 200, capi.RT_TYPE_RGB),
    ("cornell_knot", dict(nu=2048, nv=213), 960, 540, capi.RT_TYPE_RGB),
    ("knot_glass", dict(nu=256, nv=32), 320, 200, capi.RT_TYPE_SPECTRAL),
    ("interior", dict(nu=96, nv=24, tex_size=64), 320, 200, capi.RT_TYPE_RGB),
    ("mixed", {}, 200, 160, capi.RT_TYPE_RGB),
    ("textured_room", {}, 200, 160, capi.RT_TYPE_RGB),
    ("fuzz", dict(seed=7), 160, 120, capi.RT_TYPE_RGB),
    ("fuzz", dict(seed=11), 160, 120, capi.RT_TYPE_SPECTRAL),
]

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 1: derived from the uploaded trees, 2: the library's own trees
bad = 0
for name, kw, W, H, rtype in CASES:
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=4)
    out = {}
    for wide in (0, MODE):
        be = capi.Backend(0)
        be.set_option(capi.RT_OPT_WIDE_BVH, wide)
        be.upload_scene(scene)
        be.resize(W, H)
        be.set_frame(cam, scene.sky, rtype)
        prim = be.trace_primary(frame=1)
        be.set_counters(True)
        be.reset_stats()
        be.render(1, 4, 1)
        img, _ = be.readback()
        out[wide] = (prim, img, be.stats())
        be.close()
    (m0, i0, t0, u0, v0), img0, st0 = out[0]
    (m1, i1, t1, u1, v1), img1, st1 = out[MODE]
    idDiff = (m0 != m1) | (i0 != i1)
    tDiff = np.abs(t0 - t1) > 1e-5 * np.maximum(np.abs(t0), 1e-30)
    pixDiff = np.any(img0 != img1, axis=-1)
    num = np.sqrt(np.mean((img0.astype(np.float64) - img1) ** 2))
    den = np.sqrt(np.mean(img0.astype(np.float64) ** 2)) + 1e-30
    print(f"{name:14s} {W}x{H} type={rtype}: id diffs {int(idDiff.sum())}, t diffs {int(tDiff.sum())}, differing pixels {int(pixDiff.sum())} of {W * H}, "
          f"rel RMSE {num / den:.2e}; nodes/ray {st0['node_visits'] / max(st0['rays'], 1):.1f} -> {st1['node_visits'] / max(st1['rays'], 1):.1f}, "
          f"tris/ray {st0['tri_tests'] / max(st0['rays'], 1):.1f} -> {st1['tri_tests'] / max(st1['rays'], 1):.1f}")
    if tDiff.sum() > 0 or num / den > 1e-2:
        bad += 1
sys.exit(1 if bad else 0)
