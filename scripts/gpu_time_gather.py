#!/usr/bin/env python3
"""Cost of the per-step exchange of bench.py's N > 1 runs, on a one-rank RCCL communicator (the most a one-GPU box holds):
crabml_amd.dist.all_gather_doubles (cached pinned/device buffers, all_gather_into_tensor) against the list form it
replaced (torch.tensor on the device, one output tensor per rank, cat, .cpu()).  The one-rank collective itself is a
copy; what this times is the host-side and copy overhead around it, which does not shrink with more ranks."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd.dist import all_gather_doubles  # noqa: E402


def old(partial, ws):
    mine = torch.tensor([partial], dtype=torch.float64, device="cuda")
    parts = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine)
    total = 0.0
    for p in torch.cat(parts).cpu().tolist():
        total += p
    return total


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    for name, fn in (("list form (before)", lambda x: old(x, 1)), ("all_gather_doubles (now)", lambda x: float(all_gather_doubles([x])[0, 0]))):
        for _ in range(50):
            fn(1.5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2000
        for i in range(n):
            assert fn(float(i)) == float(i)
        print(f"{name}: {1e6 * (time.perf_counter() - t0) / n:.1f} us per exchange", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
