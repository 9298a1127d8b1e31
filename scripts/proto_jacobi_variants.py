"""numpy prototypes behind kernel 0's design (round 2): one-sided (Hestenes) against two-sided cyclic Jacobi with the same
round-robin pairing, on symmetrised codon rate matrices.  The one-sided variant needs 11-14 sweeps (some omega values more)
where the two-sided one needs 8-10, and wants an absolute floor for the null column; kernel 0 is two-sided.
    python scripts/proto_jacobi_variants.py"""
import numpy as np, sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from crabml_amd import synth

def circle_pairs(step):
    # players 1..63 on a ring, player 0 fixed; same as kernel: k=0: (step, 63)->(63 is dummy) ; p=(step+k)%63, q=(step-k)%63
    ps, qs = [], []
    for k in range(32):
        if k == 0: p, q = step, 63
        else: p, q = (step + k) % 63, (step - k + 63) % 63
        ps.append(p); qs.append(q)
    return np.array(ps), np.array(qs)

def hestenes(S, max_sweeps=30):
    n = S.shape[0]
    W = np.zeros((64, 64)); W[:n, :n] = S
    V = np.eye(64)
    tol_abs = (64 * 2.0**-53) ** 2 * (S * S).sum()
    eps2 = 2.0 ** -106
    for sweep in range(max_sweeps):
        nrot = 0
        for step in range(63):
            p, q = circle_pairs(step)
            ok = (p < n) & (q < n)
            p, q = p[ok], q[ok]
            a = (W[:, p] ** 2).sum(0); b = (W[:, q] ** 2).sum(0); g = (W[:, p] * W[:, q]).sum(0)
            rot = (g * g > eps2 * a * b) & (np.minimum(a, b) > tol_abs)
            p, q, a, b, g = p[rot], q[rot], a[rot], b[rot], g[rot]
            nrot += len(p)
            d = b - a; bb = 2 * g
            h = np.sqrt(d * d + bb * bb); D = np.abs(d) + h
            r = 1 / np.sqrt(D * D + bb * bb)
            c = D * r; s = np.where(d >= 0, 1.0, -1.0) * bb * r
            wp, wq = W[:, p].copy(), W[:, q].copy()
            W[:, p] = c * wp - s * wq; W[:, q] = s * wp + c * wq
            vp, vq = V[:, p].copy(), V[:, q].copy()
            V[:, p] = c * vp - s * vq; V[:, q] = s * vp + c * vq
        if nrot == 0: break
    V = V[:n, :n]
    # Newton-Schulz
    V = V + V @ (0.5 * (np.eye(n) - V.T @ V))
    lam = np.einsum('ij,ij->j', V, S @ V)
    return lam, V, sweep + 1

def two_sided(S, max_sweeps=30):
    n = S.shape[0]
    A = np.zeros((64, 64)); A[:n, :n] = S
    V = np.eye(64); eps2 = 2.0 ** -106
    for sweep in range(max_sweeps):
        nrot = 0
        for step in range(63):
            p, q = circle_pairs(step)
            ok = (p < n) & (q < n); p, q = p[ok], q[ok]
            app, aqq, apq = A[p, p], A[q, q], A[p, q]
            rot = apq * apq > eps2 * np.abs(app * aqq)
            A[p[~rot], q[~rot]] = 0; A[q[~rot], p[~rot]] = 0
            p, q, app, aqq, apq = p[rot], q[rot], app[rot], aqq[rot], apq[rot]
            nrot += len(p)
            tau = (aqq - app) / (2 * apq); t = np.where(tau >= 0, 1.0, -1.0) / (np.abs(tau) + np.sqrt(1 + tau * tau))
            c = 1 / np.sqrt(1 + t * t); s = t * c
            for M in (A, V):
                mp_, mq = M[:, p].copy(), M[:, q].copy()
                M[:, p] = c * mp_ - s * mq; M[:, q] = s * mp_ + c * mq
            ap, aq = A[p, :].copy(), A[q, :].copy()
            A[p, :] = c[:, None] * ap - s[:, None] * aq; A[q, :] = s[:, None] * ap + c[:, None] * aq
            A[p, q] = 0; A[q, p] = 0
        if nrot == 0: break
    lam = np.diag(A)[:n].copy(); V = V[:n, :n]
    V = V + V @ (0.5 * (np.eye(n) - V.T @ V))
    return lam, V, sweep + 1

def P_of(lam, V, pi, t):
    sq = np.sqrt(pi)
    return (V / sq[:, None]) @ np.diag(np.exp(lam * t)) @ (V * sq[:, None]).T

if __name__ == "__main__":
    from crabml_amd import models as m
    rng = np.random.default_rng(1)
    for trial in range(4):
        pi = rng.dirichlet(np.ones(61) * (0.3 if trial % 2 else 3.0))
        kappa = rng.uniform(0.5, 8)
        for omega in (1e-6, 1e-3, 0.05, 0.5, 1.0, 2.5, 20.0):
            Q = m.build_codon_Q_matrix(kappa, omega, pi)
            sq = np.sqrt(pi)
            S = (Q * sq[:, None]) / sq[None, :]
            S = 0.5 * (S + S.T)
            lr, Vr = np.linalg.eigh(S)
            row = []
            for name, f in (("one-sided", hestenes), ("two-sided", two_sided)):
                lam, V, sw = f(S)
                row.append(f"{name}: {sw} sweeps, residual {np.abs(S @ V - V * lam).max() / np.abs(S).max():.1e}, "
                           f"orthogonality {np.abs(V.T @ V - np.eye(61)).max():.1e}")
            print(f"kappa {kappa:.2f} omega {omega:g}: " + "; ".join(row))
