"""Data parallelism over independent graph trajectories.

The reference has no multi-process code; a batched GNNGraph is block-diagonal
(/root/reference/test/runtests.jl:89-102, src/layers.jl:359-361), messages never cross graphs, so whole
trajectories shard across GPUs with replicated parameters and ONE all-reduce(sum) of the flat
parameter-gradient vector per backward pass (33 KB for 2 x GCNConv(64=>64): latency bound; RCCL picks
its low-latency protocol for this size on the fully connected xGMI mesh).  One process per GPU;
backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous range of whole trajectories owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten_grads(tree):
    """Flat fp32 vector of all gradient leaves of a nested dict (ComponentArray order: insertion order)."""
    leaves = []

    def walk(t):
        for v in t.values():
            if isinstance(v, dict):
                walk(v)
            else:
                leaves.append(v)
    walk(tree)
    return torch.cat([g.reshape(-1) for g in leaves]), leaves


def allreduce_gradients(tree, group=None, average=False):
    """In-place all-reduce(sum) of every gradient leaf of `tree` (a nested dict of tensors) as ONE
    collective on the flat vector; with average=True divides by the world size afterwards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tree
    flat, leaves = flatten_grads(tree)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in leaves:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return tree
