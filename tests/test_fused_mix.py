"""The end of an evaluation, three switches; each is read once per process, so the same evaluations are repeated in child
processes and compared as hexadecimal floats.
- With at most 256 patterns mix_kernel + final_sum_kernel run as ONE launch (mix_final_kernel), which must give the bits of
  those two launches (CRAB_NO_FUSED_MIX=1 with CRAB_NO_WIDE_MIX=1).
- The final-sum kernel stores the log-likelihoods straight into the engine's pinned host buffer instead of a device buffer
  that is then copied: CRAB_NO_ZERO_COPY=1 must give the same bits.
- Log-sum-exp mixtures of up to 16 classes run with one (pattern, class) per lane (mix_wide_kernel): every per-site value
  must have the bits of the one-pattern-per-lane kernel; the totals are added in another grouping and agree to rounding."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (model, taxa, codons, seed): pattern counts on both sides of one block (256)
CASES = [("M8", 6, 40, 1), ("M8", 7, 130, 8), ("M8", 30, 400, 2), ("M0", 12, 255, 3), ("M7", 10, 700, 4), ("ModelA", 6, 120, 5),
         ("ModelA", 16, 500, 9), ("M8", 40, 2600, 6), ("free-ratio", 8, 150, 7)]


def evaluate_all():
    from crabml_amd import synth
    from crabml_amd.engine import Engine
    out = []
    for model, n_taxa, n_sites, seed in CASES:
        case = synth.make_case(model, n_taxa, n_sites, seed=seed)
        with Engine(0) as eng:
            eng.set_tree(case.parent, case.leaf_node_ids)
            eng.set_alignment(case.codes, n_states=61)
            row = {"n_patterns": int(eng.n_patterns)}
            if model == "ModelA":      # MIX_LINEAR
                vals = [eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi,
                                        case.branch_lengths, case.branch_labels)]
            elif model == "free-ratio":  # one class
                vals = [eng.branch(case.Q_per_node, case.pi, case.branch_lengths)]
            else:
                lnL, sites = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
                lnL_nc = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, no_clamp=True)
                B = 5
                Q = np.repeat(np.asarray(case.Q)[None], B, axis=0)
                pr = np.repeat(np.asarray(case.proportions)[None], B, axis=0)
                bl = np.repeat(case.branch_lengths[None], B, axis=0).copy()
                for b in range(1, B):
                    bl[b, b] *= 1.0 + 1e-5
                batch = eng.site_classes_batch(Q, pr, case.pi, bl)
                vals = [lnL, lnL_nc] + list(batch) + [float(np.sum(sites))] + list(np.asarray(sites)[:16])
            row["hex"] = [float(v).hex() for v in vals]
        out.append(row)
    return out


@pytest.mark.gpu
def test_fused_mixture_launch_gives_the_bits_of_the_two_launches(gpu_engine_cls):
    assert "CRAB_NO_FUSED_MIX" not in os.environ
    fused = evaluate_all()
    assert "CRAB_NO_ZERO_COPY" not in os.environ and "CRAB_NO_WIDE_MIX" not in os.environ

    def child(**switches):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, **switches), capture_output=True,
                           text=True, timeout=60)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    def same_sites_close_totals(a, b):
        for ra, rb in zip(a, b):
            assert ra["n_patterns"] == rb["n_patterns"] and len(ra["hex"]) == len(rb["hex"])
            if len(ra["hex"]) > 1:                                  # site-class rows end with 16 per-site values
                assert ra["hex"][-16:] == rb["hex"][-16:]
            for x, y in zip(ra["hex"], rb["hex"]):
                x, y = float.fromhex(x), float.fromhex(y)
                assert abs(x - y) <= 1e-14 * max(1.0, abs(x))

    n_pat = [row["n_patterns"] for row in fused]
    assert sum(n <= 256 for n in n_pat) >= 4 and max(n_pat) > 256       # the fused form (every mixture mode) and the two launches
    assert fused == child(CRAB_NO_ZERO_COPY="1")
    assert fused == evaluate_all()              # and again in this process: the host buffer is reused from call to call
    narrow = child(CRAB_NO_WIDE_MIX="1")
    assert narrow == child(CRAB_NO_WIDE_MIX="1", CRAB_NO_FUSED_MIX="1") == child(CRAB_NO_WIDE_MIX="1", CRAB_NO_FUSED_MIX="1", CRAB_NO_ZERO_COPY="1")
    same_sites_close_totals(fused, narrow)
    same_sites_close_totals(fused, child(CRAB_NO_FUSED_MIX="1"))
    for a, b in zip(fused, narrow):             # at most 256 patterns: the fused launch either way, the same bits
        if a["n_patterns"] <= 256:
            assert a == b
    for row in fused:
        assert all(np.isfinite(float.fromhex(h)) for h in row["hex"])


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    print(json.dumps(evaluate_all()))
