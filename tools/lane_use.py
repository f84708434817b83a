#!/usr/bin/env python3
"""Where the traversal kernel's lanes go: per bounce stage of the default workload, the share of lane
slots that hold a ray and the share that does work (rt_stats with counters on).
    python tools/lane_use.py [spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = capi.Scene("cornell_knot", nu=2048, nv=213)
W, H = 1920, 1080
be = capi.Backend(0)
be.upload_scene(scene)
be.resize(W, H)
be.set_counters(True)
prev = None
for bounces in range(0, 5):
    cam = capi.make_camera(width=W, height=H, bounces=bounces, ray_per_pixel=1)
    be.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
    be.reset_stats()
    be.reset_accum()
    be.render(1, spp, 1)
    st = be.stats()
    cur = {k: st[k] for k in ("rays", "node_visits", "tri_tests", "leaf_visits", "wave_steps", "active_lane_steps", "inner_phase_steps", "refills")}
    cur["trace_ms"] = st["trace_ms"]
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    # a run with `bounces` = b adds stage b's extension rays, and the shadow rays of stages b and b+1 move
    work = d["node_visits"] + d["leaf_visits"]
    slots = 64 * max(d["wave_steps"], 1)
    print(f"bounces<={bounces}: +{d['rays'] / 1e6:7.1f} Mrays +{d['trace_ms']:7.2f} ms  nodes/ray {d['node_visits'] / max(d['rays'], 1):5.2f} leaves/ray {d['leaf_visits'] / max(d['rays'], 1):4.2f} "
          f"tris/ray {d['tri_tests'] / max(d['rays'], 1):4.2f} | lanes holding a ray {d['active_lane_steps'] / slots:5.1%}, working {work / slots:5.1%}, "
          f"inner-phase steps {d['inner_phase_steps'] / max(d['wave_steps'], 1):5.1%}, rays per refill {d['rays'] / max(d['refills'], 1):5.1f}, steps/ray {d['wave_steps'] * 64 / max(d['rays'], 1):6.1f}")
    prev = cur
be.close()
