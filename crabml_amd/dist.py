"""Multi-GPU evaluation: site patterns shard across ranks, one process per GPU.

Patterns are independent given the parameters (the reference never mixes alignment rows,
rust/src/likelihood.rs:176-195), so each rank owns a contiguous slice of the compressed pattern list and
every transition matrix; the only exchange is the partial log-likelihood: one fp64 per rank,
all-gathered over RCCL (``backend="nccl"`` on ROCm) and summed in rank order on every rank, so the total
is bit-identical on all ranks and from run to run.  With world_size 1 nothing is exchanged.

The evaluator is injected, so the same plumbing runs over gloo on CPUs in the tests.
"""
from __future__ import annotations

import numpy as np


def pattern_slices(n_patterns: int, world_size: int, align: int = 128):
    """Contiguous [begin, end) per rank, boundaries on multiples of ``align`` (the kernel's tile width)
    so no tile straddles two GPUs; the last rank takes the remainder."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    tiles = -(-n_patterns // align)
    per = -(-tiles // world_size)
    out = []
    for r in range(world_size):
        b = min(n_patterns, r * per * align)
        e = min(n_patterns, (r + 1) * per * align)
        out.append((b, e))
    return out


def ordered_sum(partial: float, group=None, device=None) -> float:
    """Sum of one double per rank, accumulated in rank order (deterministic, identical on every rank)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(partial)
    ws = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([partial], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine, group=group)
    total = 0.0
    for p in torch.cat(parts).cpu().tolist():
        total += p
    return total


class ShardedLikelihood:
    """Wraps a per-rank evaluator ``evaluate(*args, **kw) -> partial lnL over this rank's patterns``."""

    def __init__(self, evaluate, group=None, device=None):
        self.evaluate = evaluate
        self.group = group
        self.device = device

    def __call__(self, *args, **kw) -> float:
        return ordered_sum(self.evaluate(*args, **kw), self.group, self.device)


def sharded_engine(codes, parents, leaf_node_ids, rank: int, world_size: int, device: int | None = None, n_states: int = 61):
    """An Engine holding rank's slice of the alignment's patterns (needs a GPU)."""
    from .engine import Engine, compress_patterns
    from ._lib import load
    import ctypes as C
    _, w, _ = compress_patterns(codes, n_states)
    width = 128
    begin, end = pattern_slices(len(w), world_size, width)[rank]
    eng = Engine(rank if device is None else device)
    eng.set_tree(parents, leaf_node_ids)
    eng.set_alignment(codes, n_states=n_states, compress=True, pattern_begin=begin, pattern_end=end)
    return eng
