"""Time evaluations that start from model parameters (Q construction behind the boundary) against the
reference's division of labour (rate matrices built in Python before every FFI call), and batched vectors (8f.1).
    python scripts/gpu_time_params.py [n_sites_C2] [n_taxa_free] [n_sites_free]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth, models
from crabml_amd.engine import Engine, discretise_omega


def rate(fn, n=20, warm=3):
    for _ in range(warm):
        fn(0)
    t = time.perf_counter()
    for i in range(n):
        fn(i + 1)
    return n / (time.perf_counter() - t)


def main():
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    nt_free = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ns_free = int(sys.argv[3]) if len(sys.argv) > 3 else 12500
    # ---- C2: M8 ----
    case = synth.make_case("M8", 100, n_sites, seed=2)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        bl, pi = case.branch_lengths, case.pi

        def py_q(i):   # what crabML does per objective call (optimizer.py:682: model.get_Q_matrices(), then the FFI)
            m = models.M8CodonModel(kappa=case.kappa * (1 + 1e-7 * i), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=pi)
            return eng.site_classes(np.stack(m.get_Q_matrices()), m.get_site_classes()[0], pi, bl)

        def params_scipy(i):  # matrices behind the boundary, the beta quantiles still from scipy
            m = models.M8CodonModel(kappa=case.kappa * (1 + 1e-7 * i), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=pi)
            pr, om = m.get_site_classes()
            return eng.codon_site_classes(m.kappa, om, pr, pi, bl)

        def params(i):  # quantiles from crab_discretise_omega as well: nothing but five scalars starts in Python
            om = np.append(discretise_omega("beta", 0.7, 1.5, 10), 2.5)
            pr = np.append(np.full(10, 0.9 / 10), 1.0 - 0.9)
            return eng.codon_site_classes(case.kappa * (1 + 1e-7 * i), om, pr, pi, bl)

        a, b = py_q(1), params(1)
        print(f"C2 M8 100x{n_sites}: lnL from Python-built Q {a:.6f}, from parameters {b:.6f} (diff {abs(a-b):.1e})")
        print(f"  Python builds Q, crab_eval_site_classes        : {rate(py_q):8.1f} evaluations/s")
        print(f"  scipy quantiles, crab_eval_codon_site_classes  : {rate(params_scipy):8.1f} evaluations/s")
        print(f"  crab_discretise_omega + crab_eval_codon_site_classes: {rate(params):8.1f} evaluations/s")
        pr, om = models.M8CodonModel(kappa=case.kappa, p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=pi).get_site_classes()
        for B in (4, 16):
            def batch(i):
                ka = case.kappa * (1 + 1e-7 * (i * B + np.arange(B)))
                return eng.codon_site_classes(ka, np.repeat(np.array(om)[None], B, 0), np.repeat(np.array(pr)[None], B, 0), pi, np.repeat(bl[None], B, 0))
            print(f"  crab_eval_codon_site_classes, {B:2d} vectors per call: {B * rate(batch, n=8, warm=2):8.1f} evaluations/s")
    # ---- free-ratio (C4 shape, one GPU's share of the sites by default) ----
    case = synth.make_case("free-ratio", nt_free, ns_free, seed=4)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        bm = models.CodonBranchModel(case.pi, np.zeros(case.n_nodes, dtype=int), free_ratio=True, n_nodes=case.n_nodes)

        def py_q(i):   # codon_branch.py:129-161 then lib.rs:363-430: n_nodes matrices built in Python, 59 MB over the FFI at C4
            w = case.omegas.copy()
            w[i % len(w)] *= 1 + 1e-6     # a finite-difference step moves one omega
            return eng.branch(bm.get_Q_matrices(case.kappa, w), case.pi, case.branch_lengths)

        def params(i):
            w = case.omega_per_node.copy()
            w[1 + i % (len(w) - 2)] *= 1 + 1e-6
            return eng.codon_branch(case.kappa, w, case.pi, case.branch_lengths)

        print(f"free-ratio {nt_free}x{ns_free} ({case.n_nodes} nodes):")
        print(f"  Python builds {case.n_nodes} Q, crab_eval_branch        : {rate(py_q, n=5, warm=1):8.2f} evaluations/s")
        print(f"  crab_eval_codon_branch, one vector             : {rate(params, n=10, warm=2):8.2f} evaluations/s   (device {eng.last_timing()[0]:.2f} ms, host eigen {eng.last_timing()[2]:.2f} ms)")
        B = 8
        def batch(i):
            w = np.repeat(case.omega_per_node[None], B, 0)
            for b in range(B):
                w[b, 1 + (i * B + b) % (w.shape[1] - 2)] *= 1 + 1e-6
            return eng.codon_branch(np.full(B, case.kappa), w, case.pi, np.repeat(case.branch_lengths[None], B, 0))
        print(f"  crab_eval_codon_branch, {B} vectors per call      : {B * rate(batch, n=4, warm=1):8.2f} evaluations/s")
    # ---- branch-site Model A (C3 shape, one GPU's share) ----
    case = synth.make_case("ModelA", 500, 6250, seed=3)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        ma = models.BranchSiteModelA(case.pi, case.branch_labels.astype(int))

        def py_q(i):   # optimize/branch_site.py:150-170
            w0 = case.omega0 * (1 + 1e-7 * i)
            fr = ma.compute_site_class_frequencies(case.p0, case.p1)
            qb, qf = ma.compute_qfactors(case.kappa, case.p0, case.p1, w0, case.omega2)
            return eng.branch_site(np.stack(ma.unnormalised_Qs(case.kappa, w0, case.omega2)), qb, qf, w0, case.omega2, fr, case.pi,
                                   case.branch_lengths, case.branch_labels)

        def params(i):
            return eng.codon_branch_site(case.kappa, case.omega0 * (1 + 1e-7 * i), case.omega2, case.p0, case.p1, case.pi,
                                         case.branch_lengths, case.branch_labels)

        print(f"Model A 500x6250:")
        print(f"  Python builds Q and Qfactors, crab_eval_branch_site: {rate(py_q, n=10):8.1f} evaluations/s")
        print(f"  crab_eval_codon_branch_site, one vector            : {rate(params, n=10):8.1f} evaluations/s")
        B = 8
        def batch(i):
            bls = np.repeat(case.branch_lengths[None], B, 0)
            for b in range(B):
                bls[b, (i * B + b) % (bls.shape[1] - 1)] *= 1 + 1e-6
            return eng.codon_branch_site(np.full(B, case.kappa), np.full(B, case.omega0), np.full(B, case.omega2), np.full(B, case.p0),
                                         np.full(B, case.p1), case.pi, bls, case.branch_labels)
        print(f"  crab_eval_codon_branch_site, {B} vectors per call     : {B * rate(batch, n=4, warm=1):8.1f} evaluations/s")


main()
