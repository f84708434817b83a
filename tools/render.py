#!/usr/bin/env python3
"""Headless progressive render of a named scene or of .glb files to PNG / PFM.

  python tools/render.py --scene cornell_knot --spp 64 --out knot.png
  python tools/render.py --scene glb:room.glb --spectral --spp 256 --out room.png --pfm room.pfm
  python tools/render.py ... --resume state.npz --checkpoint state.npz      # continue an earlier run

Frames are numbered from 1 like the reference's frame counter, so a resumed run continues the
same sample sequence it would have produced in one go.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_knot")
    ap.add_argument("--nu", type=int, default=512)
    ap.add_argument("--nv", type=int, default=64)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--spp", type=int, default=64, help="progressive frames to add (rayPerPixel samples each)")
    ap.add_argument("--ray-per-pixel", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--spectral", action="store_true")
    ap.add_argument("--hero", action="store_true", help="Spectral with four wavelengths per camera sample (HERO_WAVELENGTH_ENABLE 1)")
    ap.add_argument("--wide", type=int, nargs="?", const=1, default=None, choices=[0, 1, 2, 3],
                    help="trees the rays walk (RT_OPT_WIDE_BVH; default: the library's, 3; 0 = the uploaded trees in the reference's order)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--pfm")
    ap.add_argument("--resume")
    ap.add_argument("--checkpoint")
    a = ap.parse_args()

    scene = capi.Scene(a.scene, nu=a.nu, nv=a.nv)
    cam = capi.make_camera(width=a.width, height=a.height, bounces=a.bounces, ray_per_pixel=a.ray_per_pixel)
    be = capi.Backend(a.device)
    if a.wide is not None:
        be.set_option(capi.RT_OPT_WIDE_BVH, a.wide)
    if a.hero:
        a.spectral = True
        be.set_option(capi.RT_OPT_HERO_WAVELENGTH, 1)
    be.upload_scene(scene)
    be.resize(a.width, a.height)
    be.set_frame(cam, scene.sky, capi.RT_TYPE_SPECTRAL if a.spectral else capi.RT_TYPE_RGB)
    done = 0
    if a.resume:
        state = np.load(a.resume)
        done = int(state["frames"])
        be.restore_accum(state["sum"], done)
    be.render(first_frame=done + 1, n_frames=a.spp, stride=1)
    total, frames = be.readback()
    capi.write_png(a.out, be.display_rgba8())
    if a.pfm:
        capi.write_pfm(a.pfm, total, 1.0 / max(frames, 1))
    if a.checkpoint:
        np.savez(a.checkpoint, sum=total, frames=frames)
    st = be.stats()
    print(f"{a.scene}: {frames} frames, {st['rays'] / 1e6:.1f} Mrays in {st['total_ms']:.1f} ms device time -> {a.out}")
    be.close()


if __name__ == "__main__":
    main()
