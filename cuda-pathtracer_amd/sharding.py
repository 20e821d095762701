"""Sharding one render over the GPUs of a node: one process per GPU, torch.distributed (RCCL).

Every (pixel, iteration, sub-sample) is independent given the counter-based RNG, so a frame
shards without any exchange during rendering; the only collective is ONE sum of the
accumulation buffers at the end (width*height*3 floats: 24.9 MB at 1080p, 99.5 MB at 4K).

* sample-range split: the render's iterations are dealt out in contiguous blocks (the first
  `iterations % world` ranks get one more), each rank renders its block into a SUM buffer; after
  the reduce the root scales by 1/iterations.  fp32 summation order differs from the sequential
  running mean, so the result matches a 1-GPU render to rounding (not bit for bit).
* tile split: the image is cut into row bands, band b belongs to rank b % world; ranks leave
  foreign pixels at zero, so the same sum reproduces the 1-GPU image bit for bit (x + 0 == x).

The functions take the renderer as a callable so that the plan and the reduce are testable on
CPU with the gloo backend; bench.py passes the HIP renderer.
"""
from __future__ import annotations

from typing import Callable, Tuple


def sample_range(rank: int, world: int, iterations: int) -> Tuple[int, int]:
    """(first_iteration, n_iterations) of `rank` out of a render of `iterations` iterations in
    total; iterations are 1-based and globally numbered.  Contiguous blocks, the first
    `iterations % world` ranks get one more (the arithmetic of MakeShardPlan in
    host/multi_gpu.hpp); a rank may get none."""
    if not (0 <= rank < world) or iterations < 0:
        raise ValueError("bad rank/world/iterations")
    base, extra = divmod(iterations, world)
    return rank * base + min(rank, extra) + 1, base + (1 if rank < extra else 0)


def owned_bands(height: int, band_rows: int, rank: int, world: int):
    """Row ranges [(y0, y1), ...] that `rank` renders under the tile split."""
    if band_rows <= 0 or band_rows % 16 != 0:
        raise ValueError("band_rows must be a positive multiple of 16")
    n_bands = (height + band_rows - 1) // band_rows
    return [(b * band_rows, min(height, (b + 1) * band_rows)) for b in range(rank, n_bands, world)]


def reduce_to_root(buffer, dist=None, dst: int = 0, async_op: bool = False, single_rank_too: bool = False):
    """Sum the accumulation buffers of all ranks into rank `dst` (in place).

    async_op=True returns the collective's work handle (or None when there is nothing to reduce):
    the reduce then runs on the backend's own stream behind the work already enqueued on the
    current stream, so the next render (into ANOTHER buffer) overlaps with it; call .wait() on
    the handle before the buffer is read or rendered into again.  single_rank_too issues the
    collective even in a group of one (a rehearsal of the RCCL path on a one-GPU box)."""
    work = None
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_too):
        work = dist.reduce(buffer, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
    return work if async_op else buffer


def render_sample_split(render_sum: Callable, rank: int, world: int, iterations: int,
                        dist=None, dst: int = 0):
    """render_sum(first_iteration, n_iterations) -> tensor holding the SUM over that range (all
    zero for an empty range).

    Returns the mean image over `iterations` iterations (the whole render's, any split over the
    ranks) on rank `dst` (other ranks get their partial sums back)."""
    first, n = sample_range(rank, world, iterations)
    buf = render_sum(first, n)
    reduce_to_root(buf, dist, dst)
    if rank == dst:
        buf.mul_(1.0 / float(iterations))
    return buf


def render_tile_split(render_bands: Callable, rank: int, world: int, dist=None, dst: int = 0):
    """render_bands(rank, world) -> tensor with this rank's bands rendered (running mean) and all
    other pixels zero.  Returns the complete image on rank `dst`."""
    buf = render_bands(rank, world)
    reduce_to_root(buf, dist, dst)
    return buf
