"""Multi-GPU decomposition of a frame: one process per GPU, disjoint sample-index ranges (and
optionally tiles) per rank, one sum-reduce of the radiance accumulators.

Every (pixel, sample) path of the reference's loop (orion/raytracer.cpp:69-88) is independent and
the only combine step is the per-pixel sum, so there is no collective on the data path itself.
The accumulator is un-normalised (rgb sums + sample count), so shards simply add.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def sample_range(total_spp: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of sample indices [0, total_spp) over `world` ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(total_spp, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def tile_grid(width: int, height: int, nx: int, ny: int) -> List[Tuple[int, int, int, int]]:
    """nx x ny half-open pixel rectangles (x0, y0, x1, y1) covering the image exactly once."""
    xs = [width * i // nx for i in range(nx + 1)]
    ys = [height * j // ny for j in range(ny + 1)]
    return [(xs[i], ys[j], xs[i + 1], ys[j + 1]) for j in range(ny) for i in range(nx)
            if xs[i + 1] > xs[i] and ys[j + 1] > ys[j]]


def shard_plan(total_spp: int, width: int, height: int, rank: int, world: int,
               tiles: Optional[Tuple[int, int]] = None):
    """Work list of one rank: [(tile or None, spp_begin, spp_end)].

    Without `tiles` each rank renders its sample range of the whole frame.  With tiles=(nx, ny)
    the (tile, sample-chunk) grid is dealt round-robin: tile t, chunk c goes to rank (t + c) % world,
    with `world` sample chunks per tile."""
    if tiles is None:
        b, e = sample_range(total_spp, rank, world)
        return [(None, b, e)] if e > b else []
    plan = []
    for t, rect in enumerate(tile_grid(width, height, *tiles)):
        for c in range(world):
            if (t + c) % world == rank:
                b, e = sample_range(total_spp, c, world)
                if e > b:
                    plan.append((rect, b, e))
    return plan


def render_plan(render_fn: Callable, plan, accum):
    """Render one rank's work list (shard_plan) into `accum`: render_fn(accum, spp_begin, spp_end, tile) ADDS samples."""
    for rect, b, e in plan:
        render_fn(accum, b, e, rect)
    return accum


def reduce_async(accum, group=None, reduce_to_all: bool = True):
    """Start the frame's one exchange, the sum of the accumulators over the process group, and return its work handle
    (None in a single process).  With the nccl backend this is one RCCL all-reduce over xGMI; the caller may render the
    next frame into another accumulator meanwhile and calls .wait() before it reads or reuses this one."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    if reduce_to_all:
        return dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=group, async_op=True)


def render_sharded(render_fn: Callable, total_spp: int, width: int, height: int, accum,
                   tiles: Optional[Tuple[int, int]] = None, group=None, reduce_to_all: bool = True):
    """Render this rank's share into `accum` (a torch tensor [H,W,4], zeroed by the caller) and sum
    the accumulators over the process group.

    render_fn(accum, spp_begin, spp_end, tile) must ADD the samples into accum (orion_amd.api.Scene.render
    does).  With the nccl backend the reduction is RCCL over xGMI; gloo works for CPU tensors."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    render_plan(render_fn, shard_plan(total_spp, width, height, rank, world, tiles), accum)
    work = reduce_async(accum, group=group, reduce_to_all=reduce_to_all)
    if work is not None:
        work.wait()
    return accum
