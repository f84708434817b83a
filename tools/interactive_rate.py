#!/usr/bin/env python3
"""Frames per second of the reference's usage pattern: one progressive frame per call
(RayEngine::render in a loop, src/main.cpp:163-202), 960x540, rayPerPixel 1, 5 bounces.
Three loops per scene: render calls only; render + the waiting display call after every frame;
render + the display call without the wait after every frame, three frames in flight."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

W, H, N, IN_FLIGHT = 960, 540, 300, 3
hiprt = C.CDLL("libamdhip64.so")


def pinned(nbytes):
    p = C.c_void_p()
    if hiprt.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), 0) != 0:
        raise RuntimeError("hipHostMalloc failed")
    return p.value


def backend(scene, rtype):
    cam = capi.make_camera(width=W, height=H, bounces=5, ray_per_pixel=1)
    be = capi.Backend(0)
    be.upload_scene(scene)
    be.resize(W, H)
    be.set_frame(cam, scene.sky, rtype)
    be.render(first_frame=1, n_frames=1, stride=1)
    be.display_rgba8_noalloc()
    be.reset_stats()
    return be


for name, kw, rtype in (("cornell_knot", dict(nu=2048, nv=213), capi.RT_TYPE_SPECTRAL), ("cornell_knot", dict(nu=2048, nv=213), capi.RT_TYPE_RGB),
                        ("roughness_demo", {}, capi.RT_TYPE_RGB)):
    scene = capi.Scene(name, **kw)

    be = backend(scene, rtype)
    t0 = time.perf_counter()
    for f in range(N):
        be.render(first_frame=2 + f, n_frames=1, stride=1)
    be.sync()
    dt = time.perf_counter() - t0
    st = be.stats()
    print(f"{name} type={rtype}: {N / dt:.0f} frames/s, {st['rays'] / dt / 1e6:.0f} Mrays/s, device {st['total_ms'] / N:.3f} ms/frame, wall {dt / N * 1e3:.3f} ms/frame", flush=True)
    be.close()

    be = backend(scene, rtype)
    t0 = time.perf_counter()
    for f in range(N):
        be.render(first_frame=2 + f, n_frames=1, stride=1)
        be.display_rgba8_noalloc()
    dt = time.perf_counter() - t0
    print(f"{name} type={rtype}, every frame displayed (rt_display_rgba8_noalloc): {N / dt:.0f} frames/s", flush=True)
    be.close()

    be = backend(scene, rtype)
    bufs = [pinned(W * H * 4) for _ in range(IN_FLIGHT)]
    t0 = time.perf_counter()
    for f in range(N):
        be.render(first_frame=2 + f, n_frames=1, stride=1)
        be.display_begin(f % IN_FLIGHT, bufs[f % IN_FLIGHT], W * H * 4)
        if f >= IN_FLIGHT - 1:
            be.display_end((f - (IN_FLIGHT - 1)) % IN_FLIGHT)
    for f in range(N - (IN_FLIGHT - 1), N):
        be.display_end(f % IN_FLIGHT)
    dt = time.perf_counter() - t0
    print(f"{name} type={rtype}, every frame displayed (rt_display_rgba8_begin/_end, {IN_FLIGHT} in flight): {N / dt:.0f} frames/s", flush=True)
    be.close()
    for b in bufs:
        hiprt.hipHostFree(C.c_void_p(b))
