#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU checker (oracle/) and the C++ host layer.

The reference renderer cannot be built or run offline and ships no fixtures, so
these vectors are outputs of the checker itself (parity unpinned by the
reference); they freeze the checker + host serialisation so that any later
change to either, or to the HIP backend, is caught.  Writes only fixtures that are missing;
name cases to regenerate them on purpose:
    python tests/golden/make_golden.py [case ...]
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import capi  # noqa: E402
from oracle_binding import OracleScene  # noqa: E402

from golden_cases import CASES, HERO_CASES, scene_name  # noqa: E402


def stream_digest(scene):
    h = hashlib.sha256()
    for slot in range(1, 6):
        h.update(np.ascontiguousarray(scene.streams[slot]).tobytes())
    return h.hexdigest()


def main():
    only = set(sys.argv[1:])   # case names; default: the cases whose fixture is missing
    glb = os.path.join(HERE, "room.glb")
    if not os.path.exists(glb):
        from glb_scenes import room_glb
        room_glb(glb, nu=32, nv=10, tex=16)
    for case, (name, kw, W, H, bounces, rpp, frames) in CASES.items():
        if (only and case not in only) or (not only and os.path.exists(os.path.join(HERE, f"{case}.npz"))):
            continue
        scene = capi.Scene(scene_name(case), **kw)
        cam = capi.make_camera(width=W, height=H, bounces=bounces, ray_per_pixel=rpp)
        orc = OracleScene(scene)
        m, i, t, u, v, cnt = orc.trace_primary(cam, frame=1)
        rgb, st = orc.render(cam, 0, 1, frames, 1)
        spec, st2 = orc.render(cam, 1, 1, frames, 1)
        np.savez_compressed(
            os.path.join(HERE, f"{case}.npz"),
            model=m, ident=i, t=t, u=u, v=v, rgb=rgb, spectral=spec,
            counts=np.array([scene.object_count, scene.models_count, scene.light_sources_count], np.int64),
            stream_words=np.array([scene.streams[s].size for s in range(1, 6)], np.int64),
            stream_sha256=np.array(stream_digest(scene)),
            rgb_stats=np.array([st[k] for k in ("rays", "node_visits", "tri_tests", "shaded_hits", "camera_samples")], np.int64),
            spectral_rays=np.array(st2["rays"], np.int64),
        )
        print(case, "written:", stream_digest(scene)[:12], st)
    for case, (name, kw, W, H, bounces, rpp, frames) in HERO_CASES.items():
        path = os.path.join(HERE, f"hero_{case}.npz")
        if (only and f"hero_{case}" not in only) or (not only and os.path.exists(path)):
            continue
        scene = capi.Scene(name, **kw)
        cam = capi.make_camera(width=W, height=H, bounces=bounces, ray_per_pixel=rpp)
        spec, st = OracleScene(scene).render(cam, 1, 1, frames, 1, hero_wavelength=True)
        np.savez_compressed(path, spectral=spec, spectral_rays=np.array(st["rays"], np.int64))
        print("hero", case, "written:", st)


if __name__ == "__main__":
    main()
