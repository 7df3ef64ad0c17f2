"""Multi-GPU: trajectories are independent units sharing theta, so an ensemble shards by contiguous
blocks of trajectories (one process per GPU) and the only exchange per gradient is one all-reduce(sum)
of [grad(np); loss] over RCCL.  The forward-only path has no collective."""
import numpy as np


def shard_bounds(n_total, world, rank):
    """contiguous block partition: first (n_total % world) ranks get one extra trajectory"""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_grad(buf, dist=None):
    """buf: torch tensor [grad(np); loss] on this rank's device (or CPU for gloo).  In place; sum over ranks."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def allreduce_counters(stats_sum, dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats_sum, op=dist.ReduceOp.SUM)
    return stats_sum


def sharded_loss_grad(local_fn, n_total, world, rank, dist=None):
    """local_fn(lo, hi) -> torch tensor [grad; loss] for trajectories [lo, hi); returns the all-reduced tensor."""
    lo, hi = shard_bounds(n_total, world, rank)
    return allreduce_grad(local_fn(lo, hi), dist)
