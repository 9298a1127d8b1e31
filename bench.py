#!/usr/bin/env python3
"""Benchmark of the likelihood hot path: BASELINE.json configs[1], M8 (10 beta + 1 omega>1 class) on a
synthetic 100-taxon x 10 000-codon alignment per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one complete lnL evaluation through the C ABI (crab_eval_site_classes): host eigendecomposition
of the 11 rate matrices, upload, batched P(t), pruning, mixture, 8-byte result back.  The alignment is
already resident in HBM (uploaded once by crab_set_alignment, outside the timed region).
Multi-GPU is weak scaling: rank r owns an independent 10 000-codon block evolved on the same tree, i.e.
the site patterns of an N x 10 000-codon alignment are sharded over N GPUs; every step ends with the
rank-ordered sum of the N partial lnL values over RCCL.

Rank 0 prints ONE JSON line.  `value` is pattern x branch x class conditional-likelihood updates per
second over all GPUs (the unit BASELINE.md uses: 21.67 M per evaluation of this workload), lnL
evaluations per second are reported beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_UPDATE = 2 * 61 * 61          # one dense CL update, 61 states, padding never counted
FP64_MFMA_PEAK_TFLOPS = 78.6           # MI355X FP64 matrix peak (AMD CDNA4 datasheet; the microarch guide lists no f64 row)
N_TAXA, N_SITES, SEED = 100, 10000, 2


def pmc_traffic():
    """HBM-side bytes per launch of the pruning kernel, from the committed rocprofv3 --pmc passes over this same
    workload (profiles/rN_pmc_c2.json, newest round: FETCH_SIZE doubled per the gfx950 correction, plus WRITE_SIZE).  A counter
    run cannot share a process with the timed run, so this is a recorded figure, not a live one; None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_c2.json")), reverse=True):   # newest round first
        try:
            with open(path) as fh:
                return float(json.load(fh)["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(case, sample_sites=10000, reps=5):
    """The oracle (CPU restatement of the Rust path, reference mode) on a bounded sample of the same workload,
    one thread per site class like the reference's Rayon loop (likelihood.rs:267)."""
    from oracle import oracle as orc
    sample_sites = min(sample_sites, case.n_sites)      # about 10 s of CPU work on the box: the whole alignment, five times each way
    codes = np.ascontiguousarray(case.codes[:, :sample_sites])
    K = len(case.proportions)
    threads = min(K, os.cpu_count() or 1, 16)
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure, case.branch_lengths,
                                              None, codes, case.leaf_node_ids, threads=threads)
        best = min(best, time.perf_counter() - t0)
    updates = sample_sites * (case.n_nodes - 1) * K
    out = {"value": updates / best, "unit": "CL updates/s", "cores": threads, "kind": "port",
           "lnl_evals_per_s": 1.0 / (best * case.n_sites / sample_sites),
           "sample": f"{sample_sites} of {case.n_sites} codons, all {K} classes, all {case.n_nodes - 1} branches dense "
                     f"(unscaled reference mode), best of {reps}; evals/s scaled linearly to {case.n_sites} codons"}
    # an all-cores variant that also splits the sites (the reference cannot: its Rayon loop is over classes
    # only), one single-threaded oracle call per block of sites; the per-site results are independent, so the blocks'
    # work adds up to the same evaluation
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
    blocks = np.array_split(np.arange(sample_sites), cores)

    def one(cols):
        return orc.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure, case.branch_lengths,
                                                     None, np.ascontiguousarray(codes[:, cols]), case.leaf_node_ids, threads=1)
    best_all = float("inf")
    with ThreadPoolExecutor(max_workers=cores) as ex:
        for _ in range(reps):
            t0 = time.perf_counter()
            list(ex.map(one, blocks))
            best_all = min(best_all, time.perf_counter() - t0)
    out["all_cores"] = {"value": updates / best_all, "cores": cores, "host_cpus": os.cpu_count(),
                        "lnl_evals_per_s": 1.0 / (best_all * case.n_sites / sample_sites),
                        "note": "same sample split into one block of sites per core, single-threaded oracle call per block"}
    out["blas"] = cpu_baseline_blas(case, codes, threads, reps=3)
    return out


def cpu_baseline_blas(case, codes, threads, reps=3):
    """The same reference-mode algorithm with the per-edge product done by the BLAS numpy ships (OpenBLAS dgemm, which
    is what crabML's ndarray `dot` calls, likelihood.rs:189): one thread per class, BLAS pinned to one thread per call.
    Of the two CPU legs this is the one closer to crabML's Rust + OpenBLAS path; the hand-loop port stays beside it."""
    from oracle import blas_baseline as bb
    sample_sites = codes.shape[1]
    K = len(case.proportions)
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        bb.site_class_log_likelihood(case.Q, case.proportions, case.pi, case.parent, case.branch_lengths, codes, threads=threads)
        best = min(best, time.perf_counter() - t0)
    updates = sample_sites * (case.n_nodes - 1) * K
    return {"value": updates / best, "unit": "CL updates/s", "cores": threads, "kind": "port",
            "lnl_evals_per_s": 1.0 / (best * case.n_sites / sample_sites), "blas": bb.blas_vendor(),
            "gflops": updates * FLOP_PER_UPDATE / best / 1e9,
            "note": f"numpy restatement of likelihood.rs:176-195 with dgemm per child edge (M = {sample_sites}, N = K = 61), "
                    f"{sample_sites} codons, best of {reps}"}


def target_workload(device, with_cpu=True, evals=8, cpu_sample=1000):
    """BASELINE.json north_star's target: M8 on a 500-taxon x 50 000-codon alignment (here: the whole alignment on ONE
    MI355X; an 8-GPU run shards its patterns).  Reported beside the headline line, never as `value`."""
    from crabml_amd import models, synth
    from crabml_amd.engine import Engine
    n_taxa, n_sites = 500, 50000
    case = synth.make_case("M8", n_taxa, n_sites, seed=6)
    K = len(case.proportions)
    Qs = []
    for i in range(evals + 2):     # a new kappa per evaluation: the eigensystem pool must not be what is timed
        m = models.M8CodonModel(kappa=case.kappa * (1.0 + 1e-7 * i), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=case.pi)
        Qs.append(np.ascontiguousarray(np.stack(m.get_Q_matrices())))
    with Engine(device) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        for Q in Qs[:2]:
            lnL = eng.site_classes(Q, case.proportions, case.pi, case.branch_lengths)
        prune = []
        t0 = time.perf_counter()
        for Q in Qs[2:]:
            lnL = eng.site_classes(Q, case.proportions, case.pi, case.branch_lengths)
            prune.append(eng.last_timing()[1])
        dt = (time.perf_counter() - t0) / evals
        dense, gather = eng.last_work()
        n_clades, sub_patterns = eng.last_clade_tables()
        n_pat = eng.n_patterns
    kernel_ms = float(np.mean(prune))
    tflops = dense * FLOP_PER_UPDATE / (kernel_ms * 1e-3) / 1e12
    updates_per_eval = n_sites * (case.n_nodes - 1) * K
    out = {"workload": "north_star target: M8 (11 classes), 500 taxa x 50000 codons, whole alignment on one MI355X, tree length "
                       "%.2f subst/codon" % case.tree_length,
           "lnl_evals_per_s": 1.0 / dt, "ms_per_eval": 1e3 * dt, "cl_updates_per_s": updates_per_eval / dt, "patterns": n_pat,
           "prune_kernel_ms": kernel_ms, "prune_tflops": tflops, "roofline_frac": tflops / FP64_MFMA_PEAK_TFLOPS, "lnL": lnL,
           "executed_dense_updates_per_eval": dense, "whole_walk_dense_updates_per_eval": n_pat * K * (case.n_nodes - 1 - n_taxa),
           "clades": n_clades, "distinct_sub_patterns": sub_patterns}
    if with_cpu:
        from oracle import oracle as orc
        from oracle import blas_baseline as bb
        cols = np.linspace(0, n_sites - 1, cpu_sample).astype(int)
        codes = np.ascontiguousarray(case.codes[:, cols])
        threads = min(K, os.cpu_count() or 1, 16)
        best_port = best_blas = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            orc.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure, case.branch_lengths,
                                                  None, codes, case.leaf_node_ids, threads=threads)
            best_port = min(best_port, time.perf_counter() - t0)
            t0 = time.perf_counter()
            bb.site_class_log_likelihood(case.Q, case.proportions, case.pi, case.parent, case.branch_lengths, codes, threads=threads)
            best_blas = min(best_blas, time.perf_counter() - t0)
        scale = n_sites / cpu_sample
        out["cpu_baseline"] = {
            "cores": threads, "kind": "port", "blas": bb.blas_vendor(),
            "lnl_evals_per_s_port": 1.0 / (best_port * scale), "lnl_evals_per_s_blas": 1.0 / (best_blas * scale),
            "sample": f"{cpu_sample} of {n_sites} codons (every {n_sites // cpu_sample}th column), all {K} classes, all "
                      f"{case.n_nodes - 1} branches dense, best of 3; evals/s scaled linearly in the number of codons"}
        out["gpu_over_cpu_one_gpu"] = {"vs_port": (1.0 / dt) * best_port * scale, "vs_blas": (1.0 / dt) * best_blas * scale}
    return out


def target_workload_sharded(rank, world, local_rank, data_group, evals=8):
    """The north_star target as the north_star states it: M8, 500 taxa x 50 000 codons, the site patterns of ONE alignment
    sharded over the N GPUs of this run (strong scaling; rank r owns slice r of the pattern list in the engine's shard order,
    every evaluation ends with the rank-ordered sum of the N partial log-likelihoods).  Every rank returns; rank 0's dict is
    the one reported."""
    import torch
    import torch.distributed as dist
    from crabml_amd import models, synth
    from crabml_amd.dist import all_gather_doubles, ordered_sum, sharded_engine
    n_taxa, n_sites = 500, 50000
    case = synth.make_case("M8", n_taxa, n_sites, seed=6)
    K = len(case.proportions)
    Qs = []
    for i in range(evals + 2):
        m = models.M8CodonModel(kappa=case.kappa * (1.0 + 1e-7 * i), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=case.pi)
        Qs.append(np.ascontiguousarray(np.stack(m.get_Q_matrices())))
    eng = sharded_engine(case.codes, case.parent, case.leaf_node_ids, rank, world, device=local_rank)
    try:
        for Q in Qs[:2]:
            lnL = ordered_sum(eng.site_classes(Q, case.proportions, case.pi, case.branch_lengths), group=data_group)
        prune = []
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for Q in Qs[2:]:
            lnL = ordered_sum(eng.site_classes(Q, case.proportions, case.pi, case.branch_lengths), group=data_group)
            prune.append(eng.last_timing()[1])
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item()) / evals
        dense, _ = eng.last_work()
        per_rank = all_gather_doubles([float(np.mean(prune)), float(dense), float(eng.n_patterns)], group=data_group)
    finally:
        eng.close()
    updates_per_eval = n_sites * (case.n_nodes - 1) * K
    return {"workload": "north_star target: M8 (11 classes), 500 taxa x 50000 codons, site patterns of ONE alignment sharded over "
                        f"{world} MI355X (strong scaling), tree length %.2f subst/codon" % case.tree_length,
            "n_gpus": world, "scaling": "strong", "lnl_evals_per_s": 1.0 / dt, "ms_per_eval": 1e3 * dt,
            "cl_updates_per_s": updates_per_eval / dt, "lnL": lnL,
            "prune_kernel_ms_per_rank": [float(x) for x in per_rank[:, 0]],
            "executed_dense_updates_per_rank": [int(x) for x in per_rank[:, 1]],
            "patterns_per_rank": [int(x) for x in per_rank[:, 2]],
            "note": "patterns are numbered in shard order (consensus codon of the column, leaves that differ from it), so that a "
                    "slice shows about 1/N of every clade's distinct sub-patterns; one all-gather of one double per rank and evaluation"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target", action="store_true", help="skip the north_star target workload block (M8, 500 taxa x 50 000 codons)")
    ap.add_argument("--fd-batch", type=int, default=0,
                    help="also time crab_eval_site_classes_batch with this many perturbed vectors per call (auxiliary line; "
                         "off by default so that a rocprofv3 --stats run of the default command averages single evaluations only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # The contract is ONE JSON line on stdout.  gloo and RCCL write banners to file descriptor 1 when a communicator comes up
    # ("[Gloo] Rank 0 is connected ...", "RCCL version : ..."): everything but the JSON line goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from crabml_amd import synth
    from crabml_amd.dist import all_gather_doubles, ordered_sum
    from crabml_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (never used by the driver): CRAB_BENCH_BACKEND=gloo CRAB_BENCH_SHARE_GPU=1 lets several
    # ranks share cuda:0 and exchange over gloo, so the multi-rank control flow can be exercised without an 8-GPU node.
    backend = os.environ.get("CRAB_BENCH_BACKEND", "nccl")
    if os.environ.get("CRAB_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    # CRAB_BENCH_FORCE_DIST=1 (rehearsal too): a ONE-rank run under torch.distributed.run goes through the whole N > 1 set-up
    # (gloo control group, RCCL data group, probe, agreement) and exchanges its partial sum over RCCL every step.
    force_dist = os.environ.get("CRAB_BENCH_FORCE_DIST") == "1" and world == 1
    torch.cuda.set_device(local_rank)
    data_group = None
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # The ranks must agree on the backend: a rank that fell back to gloo on its own while the others sit in an
            # RCCL collective would hang the job.  The default group is gloo (control plane: barriers, the agreement
            # below); the data path is an RCCL group created beside it and probed once, and the probe's outcome is
            # agreed over gloo (MIN over ranks): either every rank exchanges over RCCL, or every rank exchanges over
            # gloo and the line says so in config.collective (one double per rank and step either way).
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ok = 1
            try:
                data_group = dist.new_group(backend="nccl")
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe, group=data_group)   # RCCL communicators are created lazily: fail here, not in the timed loop
                torch.cuda.synchronize()
            except Exception as exc:
                print(f"[bench rank {rank}] RCCL unavailable ({type(exc).__name__}: {exc})", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rank == 0:
                    print("[bench] RCCL could not be initialised on every rank: ALL ranks exchange over gloo (reported as such)",
                          file=sys.stderr, flush=True)
                data_group, backend = None, "gloo"
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    case = synth.make_case("M8", N_TAXA, N_SITES, seed=SEED, block=rank)
    eng = Engine(local_rank)
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    K = len(case.proportions)
    n_branches = case.n_nodes - 1

    # Every step gets its own parameter vector (kappa nudged by 1e-7 per step, Q stacks built here, outside the timed
    # region, as crabML's Python model classes build them outside the boundary): an optimiser never evaluates the same
    # rate matrices twice in a row, so the engine's eigensystem cache must not be what is timed.
    from crabml_amd import models
    n_q = args.warmup + args.steps
    Qs = []
    for i in range(n_q):
        m = models.M8CodonModel(kappa=case.kappa * (1.0 + 1e-7 * i), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=case.pi)
        Qs.append(np.ascontiguousarray(np.stack(m.get_Q_matrices())))
    it = iter(Qs)

    def step():
        partial = eng.site_classes(next(it), case.proportions, case.pi, case.branch_lengths)
        if force_dist:            # ordered_sum short-cuts on one rank: call the collective it is built on
            return float(all_gather_doubles([partial], group=data_group)[:, 0].sum())
        return ordered_sum(partial, group=data_group)

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lnL = step()
    prune_ms, device_ms, eigen_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnL = step()
        d, p, e = eng.last_timing()
        device_ms.append(d)
        prune_ms.append(p)
        eigen_ms.append(e)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # default (gloo) group
        elapsed = float(t.item())

    dense, gather = eng.last_work()
    n_clades, sub_patterns = eng.last_clade_tables()
    launches = eng.last_prune_launches()
    n_pat = eng.n_patterns
    # Outside the timed region, rank 0 at N = 1: the same evaluations as whole walks (clade tables off) -- the kernel's rate on
    # one product per (pattern, internal branch, class), the figure of rounds 1 and 2a, for the reader who wants both.
    whole = None
    if world == 1 and n_clades > 0 and not args.no_cpu_baseline:
        eng.set_clade_tables(False)
        ww = []
        for i in range(12):
            eng.site_classes(Qs[i % len(Qs)] * (1.0 + 1e-9 * (i + 1)), case.proportions, case.pi, case.branch_lengths)
            if i >= 2:
                ww.append(eng.last_timing()[1])
        wd = eng.last_work()[0]
        eng.set_clade_tables(True)
        w_ms = float(np.mean(ww))
        w_tf = wd * FLOP_PER_UPDATE / (w_ms * 1e-3) / 1e12
        whole = {"kernel_ms": w_ms, "dense_updates_per_launch": wd, "achieved": w_tf, "frac": w_tf / FP64_MFMA_PEAK_TFLOPS,
                 "note": "CRAB_NO_CLADE_TABLES: one launch, one product per (pattern, internal branch, class); 10 evaluations"}
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        updates_per_eval = world * N_SITES * n_branches * K           # reference units: every site, every branch, every class
        # With clade tables (subtree site repeats) a step launches the pruning kernel several times -- clade programs over the
        # clades' distinct sub-patterns in one launch, then the main programs -- and executes fewer products than
        # one per (pattern, internal branch, class).  The roofline counts what is EXECUTED, per launch, against the mean launch duration (what rocprofv3's kernel
        # stats average); the nominal count of a whole walk stands beside it.
        kernel_ms_step = float(np.mean(prune_ms))
        kernel_ms = kernel_ms_step / launches
        achieved = dense * FLOP_PER_UPDATE / (kernel_ms_step * 1e-3) / 1e12
        nominal_dense = n_pat * K * (n_branches - N_TAXA)             # products of a whole walk: internal branches only (leaves are gathers)
        out = {
            "metric": "pattern x branch CL updates/sec, M8 (lnL evals/sec alongside)",
            "value": updates_per_eval * args.steps / elapsed,
            "unit": "CL updates/s",
            "lnl_evals_per_s": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: M8 (10 beta + 1 omega>1 class), 100 taxa x 10000 codons per GPU, "
                                   "random bushy tree, tree length %.2f subst/codon, 1%% gap codons" % case.tree_length,
                       "taxa": N_TAXA, "codons_per_gpu": N_SITES, "patterns_per_gpu": n_pat, "classes": K,
                       "branches": n_branches, "sharding": f"patterns over {world} GPU(s)",
                       "collective": "none" if world == 1 and not force_dist else ("rccl" if backend == "nccl" else backend),
                       "lnL": lnL},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(),
                         "kernel": "prune_kernel_nt1", "kernel_ms": kernel_ms, "launches_per_step": launches,
                         "algorithmic_flop_per_launch": dense * FLOP_PER_UPDATE / launches,
                         "dense_updates_per_launch": dense / launches, "gathered_leaf_updates_per_launch": gather / launches,
                         "executed_dense_updates_per_step": dense, "whole_walk_dense_updates_per_step": nominal_dense,
                         "clade_tables": {"clades": n_clades, "distinct_sub_patterns": sub_patterns,
                                          "note": "clades of <= 64 leaves are walked once per distinct sub-pattern, inner clades first (one "
                                                  "launch: a program waits for the tables it gathers from), the main programs gather "
                                                  "their messages (second launch); "
                                                  "results bit-identical to whole walks (tests/test_clade_tables.py); "
                                                  "CRAB_NO_CLADE_TABLES=1 restores whole walks"},
                         "reference_algorithm_equivalent": {
                             "achieved": nominal_dense * FLOP_PER_UPDATE / (kernel_ms_step * 1e-3) / 1e12,
                             "frac": nominal_dense * FLOP_PER_UPDATE / (kernel_ms_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "note": "the products of the reference's algorithm (one per pattern, internal branch and class: SURVEY 8(d)'s "
                                     "unit) over the same kernel time; above 1 only because most of them are not executed -- the kernel's own "
                                     "rate is `frac` above"},
                         "whole_walks": whole},
            "breakdown_ms": {"device_section": float(np.mean(device_ms)), "prune_kernel": kernel_ms_step,
                             "host_eigen": float(np.mean(eigen_ms))},
        }
        if world == 1 and args.fd_batch > 0:
            # the same hot path used the way a finite-difference gradient uses it: B parameter vectors
            # per launch sequence, branch lengths perturbed one at a time, through crab_eval_site_classes_batch
            B = args.fd_batch
            Qb = np.repeat(Qs[-1][None], B, axis=0)
            prb = np.repeat(np.asarray(case.proportions)[None], B, axis=0)
            blb = np.repeat(case.branch_lengths[None], B, axis=0).copy()
            for b in range(1, B):
                blb[b, b] *= 1.0 + 1e-6
            eng.site_classes_batch(Qb, prb, case.pi, blb)
            tb = time.perf_counter()
            reps = 5
            for _ in range(reps):
                eng.site_classes_batch(Qb, prb, case.pi, blb)
            out["batched_fd"] = {"lnl_evals_per_s": B * reps / (time.perf_counter() - tb), "batch": B,
                                 "note": "auxiliary, not the headline: perturbed branch-length vectors, one call per batch"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(case)
    eng.close()
    if world > 1 and not args.no_target:
        # the north_star number itself: ONE 500 x 50 000 alignment sharded over the ranks of this run (all ranks take part)
        tw = target_workload_sharded(rank, world, local_rank, data_group)
        if rank == 0:
            out["target_workload"] = tw
    if rank == 0:
        if world == 1 and not args.no_target:
            out["target_workload"] = target_workload(local_rank, with_cpu=not args.no_cpu_baseline)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
