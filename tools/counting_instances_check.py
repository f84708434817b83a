"""Renders small scenes with the visit counters on (the counting instances of the traversal kernels)
on the default four-wide trees and compares image and ray count with the CPU checker.

Used to confirm the cause of wrong images from the four-wide alpha-cut counting instances when they are
compiled at the plain instances' register budget (-DRT_WCOUNT=0): a hand-written v_max_f32 read a
v_rcp_f32 result in the next instruction.  Prints one line per scene; exit status 1 on any difference.
"""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from raytracer_amd import capi          # noqa: E402
from oracle_binding import OracleScene  # noqa: E402

SCENES = [("textured_room", dict(nu=64, nv=16, tex_size=32)),      # alpha-cut textures + analytic primitives
          ("interior", dict(nu=48, nv=12, tex_size=32)),           # alpha-cut textures, six models (root table in LDS)
          ("mixed", dict(nu=96, nv=16)),
          ("cornell_knot", dict(nu=128, nv=24))]

bad_total = 0
for name, kw in SCENES:
    scene = capi.Scene(name, **kw)
    cam = capi.make_camera(width=48, height=40, bounces=3, ray_per_pixel=1)
    want, ost = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    b = capi.Backend(0)
    b.set_option(capi.RT_OPT_WIDE_BVH, 3)
    b.upload_scene(scene)
    b.resize(48, 40)
    b.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 0)
    b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    b.set_counters(True)
    b.reset_stats()
    b.render(1, 2, 1)
    got, _ = b.readback()
    st = b.stats()
    b.close()
    bad = int(np.any(got != want, axis=-1).sum())
    bad_total += bad + (st["rays"] != ost["rays"])
    print(f"{name}: {bad} differing pixels, rays {st['rays']} (checker {ost['rays']}), shaded hits {st['shaded_hits']} ({ost['shaded_hits']})", flush=True)
sys.exit(1 if bad_total else 0)
