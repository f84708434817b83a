#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-traced hot path on MI355X.

Workload (the headline configuration): dragon-class diffuse mesh (procedural
knot tube, 872,448 triangles) inside a Cornell box, 1920x1080, 64 samples per
pixel (64 progressive frames x 1 sample), primary + 4 bounces, RGB integrator
with NEE + MIS.  One *step* = those 64 spp over the whole image on every GPU.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The scene
is replicated; rank r renders frames r+1, r+1+N, ... (sample-split, weak
scaling: 64 spp per GPU per step) into its own fp32 sum buffer, and the buffers
are summed onto rank 0 with one RCCL reduce per step inside the timed region.
Either a launcher provides the ranks (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the
environment), or, when WORLD_SIZE is not set, `python bench.py --gpus N` starts
the N ranks itself as child processes (the parent never touches a GPU).

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
TREES = {   # RT_OPT_WIDE_BVH
    0: "uploaded trees walked in the reference's order",
    1: "four-wide trees collapsed from the uploaded trees (closest hits; the other triangle of a distance tie may win)",
    2: "the library's four-wide SAH trees (closest hits; the other triangle of a distance tie may win)",
    3: "the library's four-wide SAH trees, rays that met a distance tie traced again in the reference's order (library default)",
    4: "the library's eight-wide SAH trees, rays that met a distance tie traced again in the reference's order",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--nu", type=int, default=2048)
    ap.add_argument("--nv", type=int, default=213)
    ap.add_argument("--tex-size", type=int, default=0, help="texture edge of the textured scenes (0 = scene default)")
    ap.add_argument("--bvh", choices=["reference", "sah"], default="reference",
                    help="tree builder: the reference's median split (default; the only mode whose traversal is the reference's) or the opt-in SAH build")
    ap.add_argument("--wide", type=int, nargs="?", const=1, default=None, choices=[0, 1, 2, 3, 4],
                    help="trees the rays walk (RT_OPT_WIDE_BVH; default: the library's, 3): 0 = the uploaded trees in the reference's visiting order, "
                         "1 = four-wide trees from the uploaded trees, 2 = from the library's own SAH trees (1 and 2: closest hits, but the other triangle of a distance tie may win), "
                         "3 = as 2 with rays that met a distance tie traced again in the reference's order")
    ap.add_argument("--scene", default="cornell_knot")
    ap.add_argument("--spectral", action="store_true")
    ap.add_argument("--hero", action="store_true", help="Spectral with four wavelengths per camera sample (the reference's HERO_WAVELENGTH_ENABLE 1)")
    ap.add_argument("--shard", choices=["sample", "tile"], default="sample",
                    help="N>1: sample-split (frames dealt round-robin) or tile-split (8x8 tiles dealt round-robin)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend; gloo + --one-device rehearses the N>1 path on a single GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optin", action="store_true", help="skip the informational figures for the other settings of RT_OPT_WIDE_BVH")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--of-spp", type=int, default=0, help="the configuration's full sample count when --spp is a bounded step of it (stated in config.workload; rates are per sample)")
    ap.add_argument("--dump-accum", default=None, help="rank 0 saves the merged fp32 sum buffer of the timed steps to this .npy file")
    return ap.parse_args()


def launch_ranks(args):
    """`--gpus N` with no launcher around: start the N ranks of this script as child processes and
    wait for them.  Runs before anything in this process has touched a GPU (torch is not even
    imported yet); rank 0's JSON line goes to this process's stdout."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes here
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    give_up_at = None
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                # A rank that died leaves the others waiting in a collective: they get a few seconds to end by
                # themselves (and say why), then they are ended.
                give_up_at = time.time() + 5.0
        if give_up_at is not None and time.time() > give_up_at:
            for q in pending:
                q.terminate()
            give_up_at = time.time() + 3600.0
        time.sleep(0.05)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    from raytracer_amd import capi
    from raytracer_amd.sharding import reduce_accum

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; measuring {world}", file=sys.stderr)
    distributed = world > 1
    if distributed:
        if args.dist_backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:   # (a torch without the argument: the communicator is made at the first collective)
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(args.dist_backend)

    args.spectral = args.spectral or args.hero
    rtype = capi.RT_TYPE_SPECTRAL if args.spectral else capi.RT_TYPE_RGB
    t_build = time.time()
    scene = capi.Scene(args.scene, nu=args.nu, nv=args.nv, tex_size=args.tex_size, bvh=args.bvh)
    t_build = time.time() - t_build
    cam = capi.make_camera(width=args.width, height=args.height, bounces=args.bounces, ray_per_pixel=1)

    be = capi.Backend(local_rank)
    if args.wide is not None:
        be.set_option(capi.RT_OPT_WIDE_BVH, args.wide)
    if args.hero:
        be.set_option(capi.RT_OPT_HERO_WAVELENGTH, 1)
    wide = args.wide if args.wide is not None else int(os.environ.get("RT_WIDE_BVH", "3") or 3)
    t_commit = time.time()
    be.upload_scene(scene)   # copies the streams and, at commit, re-packs them and builds the library's own trees
    t_commit = time.time() - t_commit
    be.resize(args.width, args.height)
    # The sum buffer is a torch tensor so RCCL can reduce it in place.
    accum = torch.zeros(args.height, args.width, 4, dtype=torch.float32, device=f"cuda:{local_rank}")
    be.bind_accum(accum.data_ptr(), accum.numel())
    tile_split = distributed and args.shard == "tile"
    if tile_split:
        be.set_tile_shard(rank, world)
    be.set_frame(cam, scene.sky, rtype)

    # N > 1: the merge of a step's frames runs on torch's stream while the backend renders the next step.
    # Rank 0 accumulates its own frames in `accum`, as a single GPU does, and receives the other ranks'
    # in `others` (the running total of every merge so far); the other ranks render into two buffers in
    # turn, so that one can be reduced (and cleared) while the other is being rendered into.  The image is
    # accum + others on rank 0.
    others = torch.zeros_like(accum) if distributed and rank == 0 else None
    targets = [accum, torch.zeros_like(accum)] if distributed and rank != 0 else [accum]
    unmerged = []   # buffers whose frames have not been reduced yet (at most one)

    def merge_pending():
        while unmerged:
            buf = unmerged.pop()
            reduce_accum(others if rank == 0 else buf, dist, dst=0)   # (the other ranks' buffer is cleared behind the reduce)

    def step(k):
        if distributed and rank != 0:
            cur = targets[k % 2]
            be.bind_accum(cur.data_ptr(), cur.numel())
        else:
            cur = accum
        if tile_split:
            # Every rank renders all frames of the step on its own tiles (strong scaling of one image).
            be.render(first_frame=1 + k * args.spp, n_frames=args.spp, stride=1)
        else:
            # Frames of this step for this rank: sample-split over ranks, fresh frames every step.
            be.render(first_frame=1 + k * args.spp * world + rank, n_frames=args.spp, stride=world)
        if distributed:
            merge_pending()          # the previous step's frames, while this step's are being traced
            unmerged.append(cur)
        be.sync()
        if distributed:
            torch.cuda.current_stream().synchronize()

    def finish():
        """The last step's merge (nothing left to overlap it with)."""
        if distributed:
            merge_pending()
            torch.cuda.current_stream().synchronize()

    def clear():
        be.reset_accum()
        for t in targets:
            t.zero_()
        if others is not None:
            others.zero_()
        torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    be.reserve(args.spp)   # path buffers for one step: no allocation inside any render call below
    # Per-ray visit counts of this workload for the algorithmic byte model: one whole step with the
    # counters on (untimed).
    be.set_counters(True)
    be.reset_stats()
    be.render(first_frame=1, n_frames=args.spp, stride=1)
    cst = be.stats()
    be.set_counters(False)
    nodes_per_ray = cst["node_visits"] / max(cst["rays"], 1)
    tris_per_ray = cst["tri_tests"] / max(cst["rays"], 1)
    hits_per_ray = cst["shaded_hits"] / max(cst["rays"], 1)
    clear()

    for k in range(args.warmup):
        step(k)
    finish()
    clear()
    be.reset_stats()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    finish()
    barrier()
    elapsed = time.perf_counter() - t0

    st = be.stats()
    devices = [f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"]
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    if args.dump_accum and rank == 0:
        import numpy as np
        np.save(args.dump_accum, (accum + others if others is not None else accum).cpu().numpy())
    local = torch.tensor([elapsed, float(st["rays"]), float(st["camera_samples"]), st["trace_ms"], float(st["trace_launches"])],
                         dtype=torch.float64, device=f"cuda:{local_rank}")
    if distributed:
        tmax = local[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = local[1:3].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays, total_samples = float(sums[0]), float(sums[1])
    else:
        total_rays, total_samples = float(st["rays"]), float(st["camera_samples"])

    if rank == 0:
        triangles = sum(scene.model_stats(m)["triangles"] for m in range(scene.models_count))
        mrays = total_rays / elapsed / 1e6
        pixels = args.width * args.height
        spp_per_s = total_samples / pixels / elapsed
        # Algorithmic bytes per ray on the packed layout: 64 B per inner node opened, 48 B per triangle
        # tested (40 B where the four-wide trees keep their leaves as 80-byte triangle pairs: scenes
        # without alpha-cut textures), 160 B of ray/hit/state queue traffic.
        tri_bytes = 40.0 if (wide in (1, 2, 3) and args.scene not in ("interior", "textured_room")) else 48.0
        b_ray = 64.0 * nodes_per_ray + tri_bytes * tris_per_ray + 160.0
        trace_s = st["trace_ms"] / 1e3
        launches = max(st["trace_launches"], 1)
        achieved = st["rays"] * b_ray / max(trace_s, 1e-12) / 1e9
        lim = measured_limits(args)
        traffic = None
        traffic_note = "not measured for this workload"
        if lim:
            per = lim["fabric_bytes_per_launch"]
            traffic = per["fetch_corrected"] + per["write"]
            traffic_note = ("bytes per launch across the L2 -> fabric boundary: rocprofv3 FETCH_SIZE plus half of the bytes the kernel reads as wide streams (FETCH_SIZE counts those at half on gfx950, "
                            "per-lane gathers exactly: tools/microbench/fetch_calib, mix_calib), plus WRITE_SIZE; " + LIMITS_PROFILE)
        # Lanes of the traversal kernel, from the counted pass: a wave runs one phase per step.
        steps = max(cst["wave_steps"], 1)
        lanes = {
            "holding_a_ray": round(cst["active_lane_steps"] / (64.0 * steps), 4),
            "working": round((cst["node_visits"] + cst["leaf_visits"]) / (64.0 * steps), 4),
            "inner_phase_share_of_steps": round(cst["inner_phase_steps"] / steps, 4),
            "wave_steps_per_ray": round(steps / max(cst["rays"], 1), 4),
            "rays_per_refill": round(cst["rays"] / max(cst["refills"], 1), 2),
        }
        out = {
            "metric": "Mrays/s",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": dist.get_world_size() if distributed else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if tile_split else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene} ({'procedural dragon-class mesh, ' if 'knot' in args.scene else ''}{triangles} triangles in {scene.models_count} models), "
                            f"{args.width}x{args.height}, {args.spp} spp per {'image' if tile_split else 'GPU'} per step" + (f" (a step is {args.spp} of the configuration's {args.of_spp} spp; rates are per sample)" if args.of_spp else "") + f", primary+{args.bounces} bounces, "
                            f"{'Spectral' if args.spectral else 'RGB'} NEE+MIS" + (", hero-wavelength mode (4 wavelengths per camera sample; the camera ray they share is traced once and counted once)" if args.hero else "") + (", opt-in SAH trees (not the reference's traversal order)" if args.bvh == "sah" else "")
                            + (", " + TREES[wide]),
                "sharding": (("tile-split 8x8 tiles" if tile_split else "sample-split frames") + f", scene replicated, {'RCCL' if args.dist_backend == 'nccl' else args.dist_backend} sum-reduce of the fp32 accumulation buffer") if distributed else "single GPU",
                "rays_per_camera_sample": round(total_rays / max(total_samples, 1), 4),
                "shadow_rays_per_camera_sample": round(st["shadow_rays"] / max(st["camera_samples"], 1), 4),
            },
            "spp_per_s": round(spp_per_s, 3),
            "camera_samples_per_s": round(total_samples / elapsed, 1),
            "roofline": {
                "bound": "hbm",
                "kernel": "k_trace",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_frac": (round(traffic / (st["trace_ms"] / launches * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
                "traffic_note": traffic_note,
                # `frac` prices the kernel's algorithmic bytes against HBM although the caches serve most of
                # them (it can exceed 1).  The limit that binds is the one below with the largest fraction.
                "frac_note": "algorithmic bytes per launch / launch time / HBM peak; the L2s and the Infinity Cache serve most of these bytes, so this can reach or pass 1 and is no distance to a limit: `binding` and `limits` are",
                "binding": (lim["binding"] if lim else None),
                "binding_frac": (lim["limits"][lim["binding"]]["frac"] if lim else None),
                "limits": ({k: {kk: vv for kk, vv in v.items() if kk != "what"} for k, v in lim["limits"].items()} if lim else None),
                "limits_note": ("share of each machine limit the traversal kernel uses: VALU issue cycles, vector-L1 request rate, L2 read bandwidth, Infinity-Cache gather rate, HBM; "
                                "rocprofv3 counters over microbenchmark or guide peaks, " + LIMITS_PROFILE.replace(".json", ".txt")) if lim else None,
                # The counters cannot be read from inside this process: they were measured on this command by tools/roofline_limits.sh.
                # Which build they belong to, and whether it is the library running now:
                "limits_measured_on": ({"library_build_id": lim.get("library_build_id"), "git_commit": lim.get("git_commit"), "kernel": lim.get("kernel_full_name")} if lim else None),
                "limits_stale": ((lim.get("library_build_id") != capi.build_id()) if lim else None),
                "library_build_id": capi.build_id(),
                "limits_other_kernels": ({k: {kk: v.get(kk) for kk in ("ms_per_step", "launches_per_step", "valu_issue", "lane_use", "waves_per_simd", "wait_share", "issue_wait_share", "l1_hit", "l2_hit", "fabric", "hbm")}
                                          for k, v in lim.get("kernels", {}).items() if k in ("k_shade_rgb", "k_connect_rgb", "k_shade_spectral", "k_connect_spectral", "k_generate_rgb", "k_resolve_rgb")} if lim else None),
                "lanes": lanes,
                "lane_use_pmc": (lim["lane_use"] if lim else None),
                "l1_hit": (lim["l1_hit"] if lim else None),
                "l2_hit": (lim["l2_hit"] if lim else None),
                "bytes_per_ray": round(b_ray, 1),
                "nodes_per_ray": round(nodes_per_ray, 3),
                "tris_per_ray": round(tris_per_ray, 3),
                "shaded_hits_per_ray": round(hits_per_ray, 4),
                "avg_launch_ms": round(st["trace_ms"] / launches, 4),
                "launches": int(launches),
                "trace_fraction_of_step": round(st["trace_ms"] / max(st["total_ms"], 1e-9), 4),
                "trace_only_mrays_per_s": round(st["rays"] / max(trace_s, 1e-12) / 1e6, 2),
            },
            "devices": devices,
            "comm": ({"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "ranks": dist.get_world_size(),
                      "collective": "one reduce(SUM) of the fp32 sum buffer per step, %d bytes, queued behind the next step's render calls so that it runs while they are traced (the last step's is not hidden)" % (accum.numel() * 4)} if distributed else None),
            "host_scene_build_s": round(t_build, 2),
            "scene_upload_commit_s": round(t_commit, 2),
        }
        if world == 1 and wide == 3 and not args.no_optin:
            # Informational figures, outside the timed region above: the same workload on the other
            # settings of RT_OPT_WIDE_BVH.  0: the uploaded trees walked in the reference's order;
            # 1 and 2: four-wide trees (collapsed from the uploaded trees / from trees the library
            # builds itself) without the second walk for rays that met a distance tie, so the other
            # triangle of such a tie may win.  `value` is the default's (3) figure.
            out["rays_traced_again"] = int(st["retraced"])
            for mode, key in ((0, "reference_order_walk"), (1, "wide_tree_from_uploaded_no_tie_retrace"), (2, "wide_tree_no_tie_retrace")):
                be.set_option(capi.RT_OPT_WIDE_BVH, mode)
                step(0)
                be.reset_stats()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(2):
                    step(1 + k)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                wst = be.stats()
                out[key] = {"value": round(wst["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "steps": 2, "ms_per_step": round(dt / 2 * 1e3, 3),
                            "trace_only_mrays_per_s": round(wst["rays"] / max(wst["trace_ms"] / 1e3, 1e-12) / 1e6, 2)}
            be.set_option(capi.RT_OPT_WIDE_BVH, 3)
        if world == 1 and not args.no_optin:
            # Informational, never the headline: the same steps with the opt-in fast-arithmetic shading kernels
            # (RT_OPT_FAST_SHADE; radiance within the RMSE contract instead of bit-exact, traversal unchanged).
            be.set_option(capi.RT_OPT_FAST_SHADE, 1)
            step(0)
            be.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(2):
                step(1 + k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            fst = be.stats()
            be.set_option(capi.RT_OPT_FAST_SHADE, 0)
            out["fast_shade_opt_in"] = {"value": round(fst["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "steps": 2, "ms_per_step": round(dt / 2 * 1e3, 3),
                                        "spp_per_s": round(fst["camera_samples"] / pixels / dt, 3),
                                        "what": "shading and light-connection kernels built with the hardware's reciprocal / square root / transcendental instructions; "
                                                "relative RMSE against the checker 1.3 % at 64 spp on this workload (tests/test_gpu_fast_shade.py), hit ids and distances unchanged"}
        if world == 1 and not args.no_optin and wide in (2, 3):
            # Informational: the same scene with the library's trees built on the device (RT_OPT_TREE_BUILD): what the
            # second commit takes and what tracing the other trees costs.  Same image bit for bit (tests/test_gpu_tree_device.py).
            t1 = time.perf_counter()
            be.set_option(capi.RT_OPT_TREE_BUILD, capi.RT_TREE_BUILD_DEVICE)   # commits the scene again
            torch.cuda.synchronize()
            t_device_commit = time.perf_counter() - t1
            be.set_frame(cam, scene.sky, rtype)
            step(0)
            be.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(2):
                step(1 + k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            tst = be.stats()
            t1 = time.perf_counter()
            be.set_option(capi.RT_OPT_TREE_BUILD, capi.RT_TREE_BUILD_HOST)
            torch.cuda.synchronize()
            t_host_commit = time.perf_counter() - t1
            be.set_frame(cam, scene.sky, rtype)
            out["tree_build_device_opt_in"] = {"value": round(tst["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "steps": 2, "ms_per_step": round(dt / 2 * 1e3, 3),
                                               "commit_s": round(t_device_commit, 3), "commit_s_host_trees": round(t_host_commit, 3),
                                               "what": "commit = validation, re-pack, tree build and uploads of a scene whose streams are already in the context "
                                                       "(rt_upload's copy is not in it); device: Morton order + locally-ordered clustering + four-wide nodes on the GPU, "
                                                       "host: binned surface-area trees on the host threads"}
        if world == 1 and not args.no_optin:
            # (the bench context is closed first: with its stream, torch's and the three of the new context the
            # process would hold more streams than the four hardware queues, and two of the three would share one)
            be.close()
            be = None
            out["one_frame_per_call"] = one_frame_per_call(scene, rtype, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, scene, cam, rtype)
            # The other CPU baseline, the reference's own shaders on Mesa llvmpipe, cannot be produced here.
            out["llvmpipe"] = llvmpipe_status()
        print(json.dumps(out), flush=True)

    if be is not None:
        be.close()
    if distributed:
        dist.destroy_process_group()


def one_frame_per_call(scene, rtype, device):
    """Informational, outside the timed region: the reference's own usage pattern on the same scene: ONE
    progressive frame per render call (RayEngine::render in the main loop, src/main.cpp:163-202), 960x540,
    rayPerPixel 1, 5 bounces.  Such calls cannot fill the chip one at a time; the library overlaps
    consecutive calls on three streams (RT_OPT_PIPELINE) with the sum buffer updated in call order."""
    from raytracer_amd import capi

    out = {"what": "frames/s of 300 render calls of one 960x540 frame each (rayPerPixel 1, 5 bounces), synchronised once at the end"}
    cam = capi.make_camera(width=960, height=540, bounces=5, ray_per_pixel=1)
    for key, pipeline in (("frames_per_s", 1), ("frames_per_s_without_overlap", 0)):
        be = capi.Backend(device)
        try:
            be.set_option(capi.RT_OPT_PIPELINE, pipeline)
            be.upload_scene(scene)
            be.resize(960, 540)
            be.set_frame(cam, scene.sky, rtype)
            for f in range(6):
                be.render(first_frame=1 + f, n_frames=1, stride=1)
            be.sync()
            be.reset_stats()
            t0 = time.perf_counter()
            for f in range(300):
                be.render(first_frame=7 + f, n_frames=1, stride=1)
            be.sync()
            dt = time.perf_counter() - t0
            out[key] = round(300 / dt, 1)
            if pipeline:
                out["mrays_per_s"] = round(be.stats()["rays"] / dt / 1e6, 1)
        finally:
            be.close()
    # ... and with every frame tonemapped and copied to the host, as the reference's loop shows every frame:
    # the display call that waits, and the begin/end pair with three frames in flight.
    import torch
    bufs = [torch.empty((540, 960, 4), dtype=torch.uint8).pin_memory() for _ in range(3)]
    for key, split in (("frames_per_s_every_frame_displayed", True), ("frames_per_s_every_frame_displayed_waiting_call", False)):
        be = capi.Backend(device)
        try:
            be.upload_scene(scene)
            be.resize(960, 540)
            be.set_frame(cam, scene.sky, rtype)
            for f in range(6):
                be.render(first_frame=1 + f, n_frames=1, stride=1)
            be.display_rgba8_noalloc()
            t0 = time.perf_counter()
            for f in range(300):
                be.render(first_frame=7 + f, n_frames=1, stride=1)
                if split:
                    be.display_begin(f % 3, bufs[f % 3].data_ptr(), bufs[f % 3].numel())
                    if f >= 2:
                        be.display_end((f - 2) % 3)
                else:
                    be.display_rgba8_noalloc()
            if split:
                be.display_end(298 % 3)
                be.display_end(299 % 3)
            be.sync()
            out[key] = round(300 / (time.perf_counter() - t0), 1)
        finally:
            be.close()
    return out


LIMITS_PROFILE = "profiles/r03_limits.json"


def measured_limits(args):
    """What rocprofv3's counters say about the kernels of this same command (tools/roofline_limits.sh
    + tools/roofline_report.py -> profiles/r03_limits.json, stamped with the library build it was measured on).  Counters cannot be read from inside this
    process, so the figures are the ones measured on this command; they are only reported for the default
    workload they were measured on."""
    path = os.path.join(ROOT, LIMITS_PROFILE)
    default = (args.scene == "cornell_knot" and args.bvh == "reference" and args.wide is None and "RT_WIDE_BVH" not in os.environ and not args.spectral and (args.width, args.height, args.spp, args.bounces, args.nu, args.nv) == (1920, 1080, 64, 4, 2048, 213))
    if not default or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def llvmpipe_status():
    """Why the reference's GLSL cannot be timed on Mesa llvmpipe here: what this machine offers for a headless
    OpenGL 3.3 context, probed now, plus the standing refusal."""
    import ctypes

    found = []
    for lib in ("libEGL.so.1", "libOSMesa.so.8", "libOSMesa.so.6", "libGL.so.1"):
        try:
            ctypes.CDLL(lib)
            found.append(lib)
        except OSError:
            pass
    display = os.environ.get("DISPLAY") or os.environ.get("WAYLAND_DISPLAY")
    probe = (f"probed on this machine: loadable {', '.join(found) if found else 'none of libEGL / libOSMesa / libGL'}; "
             f"{'display ' + display if display else 'no DISPLAY or WAYLAND_DISPLAY'}")
    return ("unavailable (" + probe + "; and running the reference's shader text was refused in the build environment, "
            "so no harness for it ships: the CPU figure reported is the checker's, cpu_baseline)")


def host_threads():
    """Threads for CPU work on this host: every core the job may use.  A GPU box reports the whole machine's
    cores (os.cpu_count()) while a job's share is smaller and is what the *_NUM_THREADS variables carry."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        allowed = min(allowed, int(share))
    return max(1, allowed)


def cpu_baseline(args, scene, cam, rtype):
    """The CPU checker (oracle/, a port of the reference GLSL) on a bounded sample of the same workload."""
    from oracle_binding import OracleScene

    threads = host_threads()
    orc = OracleScene(scene)
    W, H = args.width, args.height
    # Calibrate on one full 1-spp frame, then size the sample (whole frames) for ~cpu-seconds of work.
    t0 = time.perf_counter()
    _, st = orc.render(cam, rtype, 1, 1, 1, threads=threads, hero_wavelength=args.hero)
    dt = time.perf_counter() - t0
    frames = int(max(1, min(64, round(args.cpu_seconds / max(dt, 1e-3)))))
    t0 = time.perf_counter()
    _, st = orc.render(cam, rtype, 1, frames, 1, threads=threads, hero_wavelength=args.hero)
    dt = time.perf_counter() - t0
    rate = st["rays"] / max(dt, 1e-9)
    y0, want_rows = 0, H
    return {
        "value": round(rate / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "cores_note": f"all cores this job may use: the machine reports {os.cpu_count()}, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}",
        "kind": "port",
        "sample": f"frames 1..{frames} of the {W}x{H} image ({frames} of the step's {args.spp} spp), {st['rays']} rays in {dt:.1f} s",
        "spp_per_s": round(st["camera_samples"] / (W * H) / max(dt, 1e-9), 5),
    }


if __name__ == "__main__":
    main()
