"""Multi-GPU evaluation: site patterns shard across ranks, one process per GPU.

Patterns are independent given the parameters (the reference never mixes alignment rows,
rust/src/likelihood.rs:176-195), so each rank owns a contiguous slice of the compressed pattern list and
every transition matrix; the only exchange is the partial log-likelihood: one fp64 per rank,
all-gathered over RCCL (``backend="nccl"`` on ROCm) and summed in rank order on every rank, so the total
is bit-identical on all ranks and from run to run.  With world_size 1 nothing is exchanged.

The evaluator is injected, so the same plumbing runs over gloo on CPUs in the tests.
"""
from __future__ import annotations

import contextlib

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


_GATHER_BUFFERS = {}   # (thread, group id, world size, device, n) -> buffers of all_gather_doubles, made once: the step loop
                       # allocates nothing.  Per thread (concurrent callers, e.g. map_genes workers, must not share a send
                       # buffer); the world size is part of the key because id() of a destroyed group can be reused.


def all_gather_doubles(vec, group=None, device=None) -> "np.ndarray":
    """ONE collective: every rank contributes ``n`` doubles and gets the [world_size, n] table, rows in rank order.
    Over RCCL the send and receive buffers live on the device and their host mirrors are pinned, all four made once per
    (group, device, n): a call is one host-to-device copy of n doubles, one ``all_gather_into_tensor`` and one copy
    back, on the current stream, with a single synchronisation at the end (no per-rank tensors, no ``cat``).  Small
    payloads only (partial sums, per-grid-point terms); big tables go through ``gather_site_table``."""
    import torch
    import torch.distributed as dist
    vec = np.ascontiguousarray(vec, dtype=np.float64).reshape(-1)
    ws = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    device = torch.device(device)
    n = vec.size
    import threading
    key = (threading.get_ident(), id(group) if group is not None else 0, ws, str(device), n)
    buf = _GATHER_BUFFERS.get(key)
    if buf is not None and buf[3].numel() != ws * n:   # (belt and braces: never reshape a stale receive buffer)
        buf = None
    if buf is None:
        on_gpu = device.type == "cuda"
        host_send = torch.empty(n, dtype=torch.float64, pin_memory=on_gpu)
        host_recv = torch.empty(ws * n, dtype=torch.float64, pin_memory=on_gpu)
        dev_send = torch.empty(n, dtype=torch.float64, device=device) if on_gpu else host_send
        dev_recv = torch.empty(ws * n, dtype=torch.float64, device=device) if on_gpu else host_recv
        if len(_GATHER_BUFFERS) >= 64:    # payload sizes are few in practice (1, B, G); do not grow without bound
            _GATHER_BUFFERS.clear()
        buf = _GATHER_BUFFERS[key] = (host_send, dev_send, dev_recv, host_recv, host_send.numpy(), host_recv.numpy().reshape(ws, n))
    host_send, dev_send, dev_recv, host_recv, send_np, recv_np = buf
    send_np[:] = vec
    if dev_send is not host_send:
        # (one process per GPU: the rank's device is the current one already, and the switch is skipped)
        here = device.index is None or torch.cuda.current_device() == device.index
        with contextlib.nullcontext() if here else torch.cuda.device(device):
            dev_send.copy_(host_send, non_blocking=True)
            dist.all_gather_into_tensor(dev_recv, dev_send, group=group)
            host_recv.copy_(dev_recv, non_blocking=True)
            torch.cuda.current_stream().synchronize()
    else:
        dist.all_gather_into_tensor(dev_recv, dev_send, group=group)
    return recv_np.copy()


def _single_rank(group) -> bool:
    import torch.distributed as dist
    return not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1


def ordered_sum(partial: float, group=None, device=None) -> float:
    """Sum of one double per rank, accumulated in rank order (deterministic, identical on every rank)."""
    if _single_rank(group):
        return float(partial)
    total = 0.0
    for p in all_gather_doubles([partial], group, device)[:, 0].tolist():
        total += p
    return total


def ordered_sum_vector(partials, group=None, device=None):
    """``ordered_sum`` for a vector: the B partial log-likelihoods of a batched evaluation (a point and its
    finite-difference neighbours) in ONE all-gather, summed in rank order on every rank."""
    partials = np.asarray(partials, dtype=np.float64)
    if _single_rank(group):
        return partials
    table = all_gather_doubles(partials, group, device)
    total = np.zeros(partials.size)
    for row in table:
        total = total + row
    return total.reshape(partials.shape)


class ShardedLikelihood:
    """Wraps a per-rank evaluator ``evaluate(*args, **kw) -> partial lnL over this rank's patterns``."""

    def __init__(self, evaluate, group=None, device=None):
        self.evaluate = evaluate
        self.group = group
        self.device = device

    def __call__(self, *args, **kw) -> float:
        return ordered_sum(self.evaluate(*args, **kw), self.group, self.device)


def sharded_engine(codes, parents, leaf_node_ids, rank: int, world_size: int, device: int | None = None, n_states: int = 61):
    """An Engine holding rank's slice of the alignment's patterns (needs a GPU).  ``device`` defaults to the
    process's LOCAL_RANK (the global rank is not a device ordinal on a multi-node job)."""
    import os
    from .engine import Engine, compress_patterns
    _, w, _ = compress_patterns(codes, n_states)
    width = 128
    begin, end = pattern_slices(len(w), world_size, width)[rank]
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", rank))
    eng = Engine(device)
    eng.set_tree(parents, leaf_node_ids)
    eng.set_alignment(codes, n_states=n_states, compress=True, pattern_begin=begin, pattern_end=end)
    return eng


# -----------------------------------------------------------------------------------------------
# [S, K] tables: compute_site_log_likelihoods_by_class needs a gather, not a reduction
# -----------------------------------------------------------------------------------------------
def gather_site_table(local: "np.ndarray", group=None, device=None) -> "np.ndarray":
    """Every rank evaluated crab_eval_site_lnl_by_class on its pattern slice: rows of ``local`` ([n_sites, K]) whose
    site belongs to another rank's slice are NaN (the engine leaves them untouched).  Returns the complete table on
    every rank: one all-gather, then each row is taken from the rank that owns it."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    ws = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(local).to(device)
    parts = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine, group=group)
    out = np.full_like(local, np.nan)
    for part in parts:
        a = part.cpu().numpy()
        have = ~np.isnan(a[:, 0])
        out[have] = a[have]
    return out


# -----------------------------------------------------------------------------------------------
# many independent genes (BASELINE config 5): shard by gene, no collective on the data path
# -----------------------------------------------------------------------------------------------
def gene_shard(n_genes: int, rank: int, world_size: int):
    """Genes of rank ``rank``: round-robin, so that sorted-by-size inputs spread evenly."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("need 0 <= rank < world_size")
    return list(range(rank, n_genes, world_size))


def map_genes(fn, genes, workers: int = 8, make_engine=None):
    """Run ``fn(engine, gene)`` for every gene on ``workers`` concurrent engines of one GPU and return the results
    in order.  One gene's evaluation fills a fraction of an MI355X (30 taxa x 400 codons x 11 classes: 77 of the 512
    resident workgroups), so several engines, each with its own stream and host thread, keep the GPU busy; the C
    library releases the GIL for the whole call.  Each worker re-targets its engine per gene (``fn`` calls
    ``set_tree`` / ``set_alignment``: microseconds for a gene), so device workspaces are per worker, not per gene."""
    from concurrent.futures import ThreadPoolExecutor
    import queue
    genes = list(genes)
    workers = max(1, min(int(workers), len(genes) or 1))
    if make_engine is None:
        from .engine import Engine
        import os
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        def make_engine():
            e = Engine(dev)
            if workers > 1 and "CRAB_DEVICE_EIGEN" not in os.environ:
                # several host threads decomposing ~40 rate matrices per gradient call each contend for the cores, the
                # GPU has room: kernel 0 takes the decompositions (profiles/r2_scan_timing.txt: 416 -> 259 ms per gene)
                e.set_eigensolver("device")
            return e
    pool = queue.Queue()
    engines = [make_engine() for _ in range(workers)]
    for e in engines:
        pool.put(e)

    def run(g):
        e = pool.get()
        try:
            return fn(e, g)
        finally:
            pool.put(e)

    try:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(run, genes))
    finally:
        for e in engines:
            close = getattr(e, "close", None)
            if close:
                close()


def sharded_beb_grid(engine, Q, pi, branch_lengths, cols, props, group=None, device=None):
    """BEB grid posteriors with the site patterns sharded over ranks: f(X | grid point) is a product over all sites, so
    the slices' per-grid-point log-likelihoods are summed across ranks (one all-gather of G doubles, added in rank
    order so that every rank holds the same bits) before each rank integrates the posteriors of its own sites.
    Returns (posteriors [n_sites, n_mix] with NaN rows for other ranks' sites, ln f(X))."""
    _, part, _ = engine.beb_grid(Q, pi, branch_lengths, cols, props, partial=True)
    if not _single_rank(group):
        total = np.zeros(part.size)
        for row in all_gather_doubles(part, group, device):
            total = total + row
        total = total.reshape(part.shape)
    else:
        total = part
    post, _, ln_fx = engine.beb_grid(Q, pi, branch_lengths, cols, props, lnfx=total)
    return post, ln_fx
