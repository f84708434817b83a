"""Frame/sample sharding of a progressive render over ranks (one process per GPU).

Sample-split: the scene is replicated and the frame numbers 1..total are dealt
round-robin, rank r taking first_frame = base + r with stride = world.  Every
frame number is rendered by exactly one rank, so the sum of the ranks' fp32
accumulation buffers equals the single-GPU sum up to addition order.
"""

from dataclasses import dataclass


@dataclass(frozen=True)
class FrameShard:
    first_frame: int  # frameCount uniform of this rank's first frame
    n_frames: int
    stride: int

    def frames(self):
        return [self.first_frame + k * self.stride for k in range(self.n_frames)]


def sample_split(total_frames: int, world: int, rank: int, base_frame: int = 1) -> FrameShard:
    """Rank `rank` of `world` renders frames base+rank, base+rank+world, ... below base+total_frames."""
    if world < 1 or not (0 <= rank < world) or total_frames < 0:
        raise ValueError("bad shard request")
    n = (total_frames - rank + world - 1) // world if total_frames > rank else 0
    return FrameShard(base_frame + rank, n, world)


def reduce_accum(accum, dist=None, dst=0, clear_others=True):
    """Sums the per-rank accumulation tensors onto `dst` in place (RCCL on GPU tensors, gloo on CPU).

    With clear_others the other ranks zero their tensor afterwards: `dst` then holds the running
    total of everything rendered so far and the others only what they rendered since the last
    merge, so the call can be repeated after every batch of frames without double counting.
    The caller must not let its renderer write into `accum` before the queued reduce/zero have
    finished (on a GPU: synchronise the stream they were issued on first).
    """
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
        if clear_others and dist.get_rank() != dst:
            accum.zero_()
    return accum
