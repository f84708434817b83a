"""N > 1 path on CPU: two gloo ranks shard the frames, render their share with the CPU
checker standing in for the device, and the reduced sum equals the single-rank sum."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.sharding import sample_split, reduce_accum


def test_sample_split_covers_every_frame_once():
    for total in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            frames = []
            for r in range(world):
                sh = sample_split(total, world, r, base_frame=5)
                assert sh.stride == world
                frames += sh.frames()
            assert sorted(frames) == list(range(5, 5 + total))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total_frames, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from raytracer_amd import capi
    from oracle_binding import OracleScene

    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    cam = capi.make_camera(width=24, height=16, bounces=3, ray_per_pixel=1)
    # Two batches, merged after each one the way bench.py's steps do: rank 0 keeps the running
    # total, the other ranks restart from zero after every merge.
    accum = None
    orc = OracleScene(scene)
    for batch in range(2):
        sh = sample_split(total_frames, world, rank, base_frame=1 + batch * total_frames)
        rgba, _ = orc.render(cam, 0, sh.first_frame, sh.n_frames, sh.stride, threads=2)
        accum = torch.from_numpy(rgba) if accum is None else accum + torch.from_numpy(rgba)
        reduce_accum(accum, dist, dst=0)
        if rank != 0:
            assert float(accum.abs().sum()) == 0.0
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_rank(tmp_path):
    total = 5
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    got = np.load(out)

    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from raytracer_amd import capi
    from oracle_binding import OracleScene

    scene = capi.Scene("cornell_knot", nu=32, nv=8)
    cam = capi.make_camera(width=24, height=16, bounces=3, ray_per_pixel=1)
    want, _ = OracleScene(scene).render(cam, 0, 1, 2 * total, 1, threads=2)
    assert np.array_equal(got[..., 3], want[..., 3])                 # every frame counted once
    assert np.allclose(got[..., :3], want[..., :3], rtol=1e-6, atol=1e-7)   # only the addition order differs


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`bench.py --gpus 2` without a launcher spawns two ranks itself; here, without a GPU, both must
    refuse to run (no CPU fallback) and the parent must hand the failure on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        return   # covered by the GPU test of the same entry point
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode != 0
    assert run.stderr.count("bench.py needs a GPU") == 2
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
