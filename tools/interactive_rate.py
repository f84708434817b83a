#!/usr/bin/env python3
"""Frames per second of the reference's usage pattern: one progressive frame per call
(RayEngine::render in a loop), 960x540, rayPerPixel 1, 5 bounces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

for name, kw, rtype in (("cornell_knot", dict(nu=2048, nv=213), capi.RT_TYPE_SPECTRAL), ("cornell_knot", dict(nu=2048, nv=213), capi.RT_TYPE_RGB),
                        ("roughness_demo", {}, capi.RT_TYPE_RGB)):
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=960, height=540, bounces=5, ray_per_pixel=1)
    be = capi.Backend(0)
    be.upload_scene(scene)
    be.resize(960, 540)
    be.set_frame(cam, scene.sky, rtype)
    be.render(first_frame=1, n_frames=1, stride=1)
    be.sync()
    be.reset_stats()
    n = 300
    t0 = time.perf_counter()
    for f in range(n):
        be.render(first_frame=2 + f, n_frames=1, stride=1)
    be.sync()
    dt = time.perf_counter() - t0
    st = be.stats()
    print(f"{name} type={rtype}: {n / dt:.0f} frames/s, {st['rays'] / dt / 1e6:.0f} Mrays/s, device {st['total_ms'] / n:.3f} ms/frame, wall {dt / n * 1e3:.3f} ms/frame")
    be.close()
