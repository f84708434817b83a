import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from raytracer_amd import capi
from oracle_binding import OracleScene
from deep_tree import with_depth
import test_gpu_deep_trees as T

for name, kw in T.SCENES[2:4]:
    for dc in ['stack24']:
        scene, orc, depth = T.deep_scene(name, kw, dc)
        cam = capi.make_camera(width=48, height=40, bounces=3, ray_per_pixel=1)
        want, ost = orc.render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
        for mode in (3,):
            for counters in (True,):
                for rep in range(3):
                    b = capi.Backend(0)
                    b.set_option(capi.RT_OPT_WIDE_BVH, mode)
                    b.upload_scene(scene); b.resize(48, 40)
                    b.set_option(capi.RT_OPT_SHADOW_EARLY_OUT, 0)
                    b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
                    b.set_counters(counters)
                    b.reset_stats()
                    b.render(1, 2, 1)
                    got, _ = b.readback()
                    st = b.stats()
                    bad = np.any(got != want, axis=-1)
                    if bad.any():
                        ys, xs = np.nonzero(bad)
                        print(f"{name} {dc} mode{mode} cnt={counters} rep{rep}: {bad.sum()} bad px rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()} zero={int((got[bad][:, :3] == 0).all(axis=1).sum())} rays {st['rays']} vs {ost['rays']}", flush=True)
                    b.close()
        print("done", name, dc, flush=True)
