#!/usr/bin/env python3
"""Config 5 scan with several processes sharing one GPU (each with its own hardware queues), `workers` engines in each.
    python scripts/gpu_scan_procs.py [n_genes] [procs] [workers]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(rank, world, n_genes, workers, ready, go, out):
    from crabml_amd import scan, synth
    genes = [synth.as_alignment_and_tree(synth.make_case("M8", 30, 400, seed=2000 + g)) for g in range(n_genes)]
    mine = genes[rank::world]
    scan.scan(genes[:2], workers=2)
    ready.put(rank)
    go.wait()
    t0 = time.perf_counter()
    rows = scan.scan(mine, null="M7", workers=workers)
    out.put((rank, time.perf_counter() - t0, [r["LRT"] for r in rows]))


if __name__ == "__main__":
    n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    ctx = mp.get_context("spawn")
    ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=child, args=(r, procs, n_genes, workers, ready, go, out)) for r in range(procs)]
    for p in ps:
        p.start()
    for _ in ps:
        ready.get(timeout=300)
    t0 = time.perf_counter()
    go.set()
    res = [out.get(timeout=600) for _ in ps]
    dt = time.perf_counter() - t0
    for p in ps:
        p.join(timeout=60)
    print(f"{procs} process(es) x {workers} worker(s): {n_genes} genes in {dt:.2f} s = {dt / n_genes * 1e3:.0f} ms per gene "
          f"(slowest process {max(r[1] for r in res):.2f} s)", flush=True)
