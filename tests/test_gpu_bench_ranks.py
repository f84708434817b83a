"""GPU: `bench.py --gpus N` starts its own ranks when no launcher did.

Rehearsal on the one GPU of the test box: two ranks and the real count of eight, all on device 0, merged
with gloo (RCCL refuses several ranks on one device); the rank logic, the frame split and the merge --
each step's reduce queued while the next step is traced -- are the ones an 8-GPU run uses.  The merged
image must be the single-context sum of the same frames."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 8])
@pytest.mark.parametrize("shard", ["sample", "tile"])
def test_bench_starts_its_ranks_and_merges_their_images(tmp_path, shard, ranks):
    W, H, spp, steps = 96, 64, 3, 3
    dump = tmp_path / "sum.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--dist-backend", "gloo", "--one-device", "--shard", shard,
                          "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--width", str(W), "--height", str(H), "--spp", str(spp),
                          "--nu", "64", "--nv", "12", "--dump-accum", str(dump)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["comm"]["ranks"] == ranks and out["comm"]["backend"] == "gloo"
    assert len(out["devices"]) == ranks and all(out["devices"][r].startswith(f"rank {r}:") for r in range(ranks))
    assert out["scaling"] == ("weak" if shard == "sample" else "strong")

    # frames of the timed steps (the warm-up step's were cleared): sample-split deals spp*ranks frames
    # per step over the ranks, tile-split renders spp frames per step on disjoint tiles
    per_step = spp * ranks if shard == "sample" else spp
    first, count = 1 + per_step, per_step * steps
    got = np.load(dump)
    scene = capi.Scene("cornell_knot", nu=64, nv=12)
    cam = capi.make_camera(width=W, height=H, bounces=4, ray_per_pixel=1)
    b = capi.Backend(0)
    try:
        b.upload_scene(scene)
        b.resize(W, H)
        b.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
        b.render(first, count, 1)
        want, n = b.readback()
    finally:
        b.close()
    assert n == count
    assert np.array_equal(got[..., 3], want[..., 3])                       # every frame counted once per pixel
    assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)   # addition order across ranks and steps differs
    if shard == "tile":
        # rank 0 keeps the running total in place: its own tiles (even row-major 8x8 tile index) are
        # accumulated frame after frame exactly as a single context does
        ys, xs = np.mgrid[0:H, 0:W]
        mine = ((ys // 8) * ((W + 7) // 8) + xs // 8) % ranks == 0
        assert np.array_equal(got[mine], want[mine])
    expected_samples = W * H * count
    assert abs(out["camera_samples_per_s"] * out["ms_per_step"] * 1e-3 * steps - expected_samples) <= 1e-3 * expected_samples
