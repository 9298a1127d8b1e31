/*
 * crab_oracle.c -- CPU restatement of crabML's Rust likelihood core.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under crabml_amd/ may include, link or call
 * this file; it is the checker (tests/, __graft_entry__.smoke) and the timed CPU
 * baseline (bench.py cpu_baseline leg), never the product.
 *
 * Each function restates one piece of /root/reference/rust/src and cites it.
 * "reference mode" = exactly the reference's algorithm: dense one-hot leaves, one
 * [S x n] * [n x n]^T product per child edge, linear-space (unscaled) conditional
 * likelihoods, clamps max(f, 1e-300) where the reference has them, none where it
 * has none (branch-site).  The only deliberate differences are the routines that
 * the reference takes from OpenBLAS/LAPACK (openblas-src 0.10.13 via
 * ndarray-linalg 0.17.0, Cargo.lock): dsyev* -> cyclic Jacobi, getrf/getri ->
 * Gauss-Jordan with partial pivoting, dgemm -> a plain blocked loop.  Only P(t) is
 * observable at the boundary, so eigenvector sign/order is free.
 *
 * Parity pin: checked against golden vectors generated from the reference's own
 * Python calculator (the .npz files under tests/golden) and against the PAML known answers the
 * reference's tests assert (tests/test_oracle.py).
 *
 * An optional "scaled" switch (per-node, per-site power-of-two rescaling with an
 * exponent accumulator) exists for inputs where the unscaled product underflows;
 * it is checked to equal reference mode wherever that is finite.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_ARG 1
#define ORC_ERR_NOROOT 2
#define ORC_ERR_SINGULAR 3
#define ORC_ERR_ALLOC 4
#define ORC_ERR_NOCONV 5

/* ------------------------------------------------------------------ */
/* symmetric eigendecomposition (stands in for LAPACK dsyev*, matrix.rs:54) */
/* ------------------------------------------------------------------ */
/* A: n x n symmetric, row-major, destroyed. evals ascending; evecs columns = vectors. */
int orc_eigh_jacobi(int n, double *A, double *evals, double *V)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            V[i * n + j] = (i == j) ? 1.0 : 0.0;
    /* the reference calls eigh(UPLO::Lower): only the lower triangle is read */
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            A[i * n + j] = A[j * n + i];
    int sweep;
    for (sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++)
                off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-34 * (diag + off) || off == 0.0)
            break;
        for (int p = 0; p < n - 1; p++) {
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0)
                    continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    if (sweep == 100)
        return ORC_ERR_NOCONV;
    for (int i = 0; i < n; i++)
        evals[i] = A[i * n + i];
    /* ascending order, as eigh returns */
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (evals[j] < evals[m])
                m = j;
        if (m != i) {
            double t = evals[i]; evals[i] = evals[m]; evals[m] = t;
            for (int k = 0; k < n; k++) {
                double v = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = v;
            }
        }
    }
    return ORC_OK;
}

/* general inverse (stands in for LAPACK getrf/getri, matrix.rs:66) */
static int invert(int n, const double *A, double *Ainv)
{
    double *M = (double *)malloc(sizeof(double) * n * 2 * n);
    if (!M) return ORC_ERR_ALLOC;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            M[i * 2 * n + j] = A[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c]))
                piv = r;
        if (M[piv * 2 * n + c] == 0.0) { free(M); return ORC_ERR_SINGULAR; }
        if (piv != c)
            for (int j = 0; j < 2 * n; j++) {
                double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t;
            }
        double d = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            Ainv[i * n + j] = M[i * 2 * n + n + j];
    free(M);
    return ORC_OK;
}

/* CachedQMatrix::new  (rust/src/matrix.rs:31-75)
 * S = Pi^1/2 Q Pi^-1/2 ; (lam, V) = eigh(S) ; U = Pi^-1/2 V ; Uinv = inverse(U). */
int orc_cached_q(int n, const double *Q, const double *pi, double *U, double *lam, double *Uinv)
{
    double *S = (double *)malloc(sizeof(double) * n * n);
    double *V = (double *)malloc(sizeof(double) * n * n);
    if (!S || !V) { free(S); free(V); return ORC_ERR_ALLOC; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            S[i * n + j] = sqrt(pi[i]) * Q[i * n + j] * (1.0 / sqrt(pi[j]));
    int rc = orc_eigh_jacobi(n, S, lam, V);
    if (rc == ORC_OK) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                U[i * n + j] = (1.0 / sqrt(pi[i])) * V[i * n + j];
        rc = invert(n, U, Uinv);
    }
    free(S); free(V);
    return rc;
}

/* CachedQMatrix::expm  (rust/src/matrix.rs:88-102): P = (U o exp(lam t)) * Uinv */
void orc_expm(int n, const double *U, const double *lam, const double *Uinv, double t, double *P)
{
    double e[256];
    double *el = (n <= 256) ? e : (double *)malloc(sizeof(double) * n);
    for (int k = 0; k < n; k++) el[k] = exp(lam[k] * t);
    for (int i = 0; i < n; i++) {
        double *row = P + (size_t)i * n;
        for (int j = 0; j < n; j++) row[j] = 0.0;
        for (int k = 0; k < n; k++) {
            double a = U[i * n + k] * el[k];
            const double *b = Uinv + (size_t)k * n;
            for (int j = 0; j < n; j++) row[j] += a * b[j];
        }
    }
    if (el != e) free(el);
}

/* matrix_exponential entry point (rust/src/lib.rs:61-79) */
int orc_matrix_exponential(int n, const double *Q, const double *pi, double t, double *P)
{
    double *U = (double *)malloc(sizeof(double) * (2 * n * n + n));
    if (!U) return ORC_ERR_ALLOC;
    double *Uinv = U + n * n, *lam = Uinv + n * n;
    int rc = orc_cached_q(n, Q, pi, U, lam, Uinv);
    if (rc == ORC_OK) orc_expm(n, U, lam, Uinv, t, P);
    free(U);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Tree::from_structure  (rust/src/likelihood.rs:36-81)                  */
/* ------------------------------------------------------------------ */
typedef struct {
    int n_nodes;
    int root;
    int *parent;      /* -1 = none */
    int *child_start; /* CSR: children of v are child[child_start[v] .. child_start[v+1]) in order of appearance */
    int *child;
    int *postorder;
    int n_post;
    char *is_leaf;    /* membership in leaf_node_ids */
    int *leaf_row;    /* node -> row of `sequences` (leaf_idx), -1 otherwise */
} orc_tree;

static void tree_free(orc_tree *t)
{
    free(t->parent); free(t->child_start); free(t->child); free(t->postorder); free(t->is_leaf); free(t->leaf_row);
}

/* node_ids[k], parents[k] (-1 = None) for k in 0..n_nodes: the (node_id, parent_id) list */
static int tree_build(orc_tree *t, int n_nodes, const int32_t *node_ids, const int32_t *parents,
                      const int32_t *leaf_node_ids, int n_leaves)
{
    memset(t, 0, sizeof(*t));
    t->n_nodes = n_nodes;
    t->parent = (int *)malloc(sizeof(int) * n_nodes);
    t->child_start = (int *)calloc(n_nodes + 1, sizeof(int));
    t->child = (int *)malloc(sizeof(int) * (n_nodes > 0 ? n_nodes : 1));
    t->postorder = (int *)malloc(sizeof(int) * n_nodes);
    t->is_leaf = (char *)calloc(n_nodes, 1);
    t->leaf_row = (int *)malloc(sizeof(int) * n_nodes);
    if (!t->parent || !t->child_start || !t->child || !t->postorder || !t->is_leaf || !t->leaf_row)
        return ORC_ERR_ALLOC;
    for (int v = 0; v < n_nodes; v++) { t->parent[v] = -1; t->leaf_row[v] = -1; }
    for (int k = 0; k < n_nodes; k++) {
        int v = node_ids ? node_ids[k] : k;
        if (v < 0 || v >= n_nodes || parents[k] >= n_nodes) return ORC_ERR_ARG;
        t->parent[v] = parents[k] < 0 ? -1 : parents[k];
    }
    /* children in order of appearance in the structure list (likelihood.rs:49-54) */
    for (int k = 0; k < n_nodes; k++)
        if (parents[k] >= 0) t->child_start[parents[k] + 1]++;
    for (int v = 0; v < n_nodes; v++) t->child_start[v + 1] += t->child_start[v];
    int *fill = (int *)calloc(n_nodes, sizeof(int));
    if (!fill) return ORC_ERR_ALLOC;
    for (int k = 0; k < n_nodes; k++) {
        int p = parents[k];
        if (p >= 0) t->child[t->child_start[p] + fill[p]++] = node_ids ? node_ids[k] : k;
    }
    free(fill);
    /* root = first node with no parent (likelihood.rs:57-59) */
    t->root = -1;
    for (int v = 0; v < n_nodes; v++)
        if (t->parent[v] < 0) { t->root = v; break; }
    if (t->root < 0) return ORC_ERR_NOROOT;
    for (int i = 0; i < n_leaves; i++) {
        int v = leaf_node_ids[i];
        if (v < 0 || v >= n_nodes) return ORC_ERR_ARG;
        t->is_leaf[v] = 1;
        t->leaf_row[v] = i;
    }
    /* post-order from the root (likelihood.rs:62-69), iterative */
    int *stack = (int *)malloc(sizeof(int) * 2 * (n_nodes + 1));
    if (!stack) return ORC_ERR_ALLOC;
    int sp = 0;
    stack[sp++] = t->root; stack[sp++] = 0;
    t->n_post = 0;
    while (sp) {
        int k = stack[sp - 1], v = stack[sp - 2];
        if (t->child_start[v] + k < t->child_start[v + 1]) {
            stack[sp - 1] = k + 1;
            int c = t->child[t->child_start[v] + k];
            if (t->n_post + sp / 2 > n_nodes) { free(stack); return ORC_ERR_ARG; } /* cycle guard */
            stack[sp++] = c; stack[sp++] = 0;
        } else {
            t->postorder[t->n_post++] = v;
            sp -= 2;
        }
    }
    free(stack);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* pruning                                                             */
/* ------------------------------------------------------------------ */
/* tmp[s][i] = sum_j L[s][j] * P[i][j]   (child_lnl.dot(&p.t()), likelihood.rs:189) */
static void gemm_nt(int S, int n, const double *L, const double *P, double *PT, double *tmp)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            PT[j * n + i] = P[i * n + j];
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        const double *l0 = L + (size_t)s * n, *l1 = l0 + n, *l2 = l1 + n, *l3 = l2 + n;
        double *o0 = tmp + (size_t)s * n, *o1 = o0 + n, *o2 = o1 + n, *o3 = o2 + n;
        for (int i = 0; i < n; i++) o0[i] = o1[i] = o2[i] = o3[i] = 0.0;
        for (int j = 0; j < n; j++) {
            const double *p = PT + (size_t)j * n;
            double a0 = l0[j], a1 = l1[j], a2 = l2[j], a3 = l3[j];
            for (int i = 0; i < n; i++) {
                double pv = p[i];
                o0[i] += a0 * pv; o1[i] += a1 * pv; o2[i] += a2 * pv; o3[i] += a3 * pv;
            }
        }
    }
    for (; s < S; s++) {
        const double *l0 = L + (size_t)s * n;
        double *o0 = tmp + (size_t)s * n;
        for (int i = 0; i < n; i++) o0[i] = 0.0;
        for (int j = 0; j < n; j++) {
            const double *p = PT + (size_t)j * n;
            double a0 = l0[j];
            for (int i = 0; i < n; i++) o0[i] += a0 * p[i];
        }
    }
}

/* leaf init (likelihood.rs:120-132): one-hot if 0 <= code < n, else all ones */
static void init_leaves(const orc_tree *t, int S, int n, const int32_t *seq, double *W)
{
    for (int v = 0; v < t->n_nodes; v++) {
        int row = t->leaf_row[v];
        if (row < 0) continue;
        double *Lv = W + (size_t)v * S * n;
        for (int s = 0; s < S; s++) {
            int32_t c = seq[(size_t)row * S + s];
            if (c >= 0 && c < n) Lv[(size_t)s * n + c] = 1.0;
            else for (int i = 0; i < n; i++) Lv[(size_t)s * n + i] = 1.0;
        }
    }
}

/* Shared body of compute_site_likelihoods (likelihood.rs:200-246) given one P per node.
 * P: [n_nodes][n][n].  site_f[s] = sum_i pi_i L_root[s][i] (unscaled mode), or in scaled
 * mode site_f[s] * 2^site_e[s].  naive != 0 selects the branch-site calculator's scalar
 * first-child-assigns loop (branch_site_likelihood.rs:274-299) instead of fill(1)+GEMM. */
static int prune(const orc_tree *t, int S, int n, const double *P, const double *pi, const int32_t *seq,
                 int naive, int scaled, double *site_f, int32_t *site_e)
{
    size_t tile = (size_t)S * n;
    double *W = (double *)calloc((size_t)t->n_nodes * tile, sizeof(double));  /* zeroed workspace, likelihood.rs:98 */
    double *tmp = (double *)malloc(sizeof(double) * tile);
    double *PT = (double *)malloc(sizeof(double) * n * n);
    int32_t *E = scaled ? (int32_t *)calloc((size_t)t->n_nodes * S, sizeof(int32_t)) : NULL;
    if (!W || !tmp || !PT || (scaled && !E)) { free(W); free(tmp); free(PT); free(E); return ORC_ERR_ALLOC; }
    init_leaves(t, S, n, seq, W);
    for (int k = 0; k < t->n_post; k++) {
        int v = t->postorder[k];
        if (t->is_leaf[v]) continue;
        double *Lv = W + (size_t)v * tile;
        int c0 = t->child_start[v], c1 = t->child_start[v + 1];
        if (!naive)
            for (size_t x = 0; x < tile; x++) Lv[x] = 1.0;  /* likelihood.rs:178 */
        for (int ci = c0; ci < c1; ci++) {
            int c = t->child[ci];
            gemm_nt(S, n, W + (size_t)c * tile, P + (size_t)c * n * n, PT, tmp);
            if (naive && ci == c0)
                memcpy(Lv, tmp, sizeof(double) * tile);      /* first child assigns */
            else
                for (size_t x = 0; x < tile; x++) Lv[x] *= tmp[x];
            if (scaled)
                for (int s = 0; s < S; s++) E[(size_t)v * S + s] += E[(size_t)c * S + s];
        }
        if (scaled) {
            for (int s = 0; s < S; s++) {
                double m = 0.0;
                for (int i = 0; i < n; i++) if (Lv[(size_t)s * n + i] > m) m = Lv[(size_t)s * n + i];
                if (m > 0.0 && m < 0x1p-64) {
                    int e; frexp(m, &e);
                    for (int i = 0; i < n; i++) Lv[(size_t)s * n + i] = ldexp(Lv[(size_t)s * n + i], -e);
                    E[(size_t)v * S + s] += e;
                }
            }
        }
    }
    int root = t->postorder[t->n_post - 1];
    const double *Lr = W + (size_t)root * tile;
    for (int s = 0; s < S; s++) {
        double f = 0.0;
        for (int i = 0; i < n; i++) f += Lr[(size_t)s * n + i] * pi[i];   /* likelihood.rs:238-242 */
        site_f[s] = f;
        if (site_e) site_e[s] = scaled ? E[(size_t)root * S + s] : 0;
    }
    free(W); free(tmp); free(PT); free(E);
    return ORC_OK;
}

static double ln_scaled(double f, int32_t e) { return log(f) + (double)e * 0.6931471805599453094; }

/* ln max(f, 1e-300) with f = fs * 2^e */
static double ln_clamped(double fs, int32_t e)
{
    if (e == 0) { double c = fs > 1e-300 ? fs : 1e-300; return log(c); }
    double l = ln_scaled(fs, e), lo = log(1e-300);
    return (fs > 0.0 && l > lo) ? l : lo;
}

/* log_sum_exp_likelihoods (likelihood.rs:349-370) */
static double log_sum_exp(int K, const double *lnf, const double *props)
{
    double mx = -INFINITY, x[1024];
    for (int k = 0; k < K; k++) { x[k] = lnf[k] + log(props[k]); if (x[k] > mx) mx = x[k]; }
    double s = 0.0;
    for (int k = 0; k < K; k++) s += exp(x[k] - mx);
    return mx + log(s);
}

/* ------------------------------------------------------------------ */
/* entry points mirroring rust/src/lib.rs                               */
/* ------------------------------------------------------------------ */

/* Per-class site likelihoods for K rate matrices sharing one tree (the par_iter bodies at
 * likelihood.rs:267-277 / :313-323).  One thread per class, as Rayon would.
 * out_f[k*S+s], out_e[k*S+s]. */
static int class_site_likelihoods(const orc_tree *t, int K, int n, int S, const double *Q, const double *pi,
                                  const double *bl, const int32_t *seq, int scaled, int threads,
                                  double *out_f, int32_t *out_e)
{
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int k = 0; k < K; k++) {
        double *U = (double *)malloc(sizeof(double) * (2 * n * n + n));
        double *P = (double *)malloc(sizeof(double) * (size_t)t->n_nodes * n * n);
        int rc = (U && P) ? ORC_OK : ORC_ERR_ALLOC;
        if (rc == ORC_OK) {
            double *Uinv = U + n * n, *lam = Uinv + n * n;
            rc = orc_cached_q(n, Q + (size_t)k * n * n, pi, U, lam, Uinv);
            if (rc == ORC_OK) {
                for (int v = 0; v < t->n_nodes; v++)   /* expm_multiple over every node, root included (likelihood.rs:207) */
                    orc_expm(n, U, lam, Uinv, bl[v], P + (size_t)v * n * n);
                rc = prune(t, S, n, P, pi, seq, 0, scaled, out_f + (size_t)k * S, out_e + (size_t)k * S);
            }
        }
        free(U); free(P);
        if (rc != ORC_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    return rc_all;
}

/* compute_site_class_log_likelihood (lib.rs:146-191 -> likelihood.rs:256-292); with K = 1 and
 * proportions = {1} this is also compute_log_likelihood (lib.rs:97-127 -> likelihood.rs:109-166):
 * ln(max(f,1e-300)) + ln 1, max + ln(exp 0) is the same number.
 * site_lnl (optional, [S]) receives the per-site log-likelihoods. */
int orc_site_class_log_likelihood(int K, int n, const double *Q, const double *props, const double *pi,
                                  int n_nodes, const int32_t *node_ids, const int32_t *parents, const double *bl,
                                  const int32_t *seq, int n_leaves, int S, const int32_t *leaf_node_ids,
                                  int scaled, int threads, double *lnL, double *site_lnl)
{
    if (K < 1 || K > 1024 || n < 1) return ORC_ERR_ARG;
    orc_tree t;
    int rc = tree_build(&t, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    double *f = (double *)malloc(sizeof(double) * (size_t)K * S);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * S);
    if (!f || !e) { free(f); free(e); tree_free(&t); return ORC_ERR_ALLOC; }
    rc = class_site_likelihoods(&t, K, n, S, Q, pi, bl, seq, scaled, threads, f, e);
    if (rc == ORC_OK) {
        double total = 0.0, lnf[1024];
        for (int s = 0; s < S; s++) {
            for (int k = 0; k < K; k++) lnf[k] = ln_clamped(f[(size_t)k * S + s], e[(size_t)k * S + s]);
            double v = log_sum_exp(K, lnf, props);
            if (site_lnl) site_lnl[s] = v;
            total += v;
        }
        *lnL = total;
    }
    free(f); free(e); tree_free(&t);
    return rc;
}

/* compute_log_likelihood proper (lib.rs:97-127 -> likelihood.rs:109-166): sum_s ln max(f_s, 1e-300) */
int orc_log_likelihood(int n, const double *Q, const double *pi, int n_nodes, const int32_t *node_ids,
                       const int32_t *parents, const double *bl, const int32_t *seq, int n_leaves, int S,
                       const int32_t *leaf_node_ids, int scaled, double *lnL, double *site_lnl)
{
    orc_tree t;
    int rc = tree_build(&t, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    double *f = (double *)malloc(sizeof(double) * (size_t)S);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
    if (!f || !e) { free(f); free(e); tree_free(&t); return ORC_ERR_ALLOC; }
    rc = class_site_likelihoods(&t, 1, n, S, Q, pi, bl, seq, scaled, 1, f, e);
    if (rc == ORC_OK) {
        double total = 0.0;
        for (int s = 0; s < S; s++) {
            double v = ln_clamped(f[s], e[s]);
            if (site_lnl) site_lnl[s] = v;
            total += v;
        }
        *lnL = total;
    }
    free(f); free(e); tree_free(&t);
    return rc;
}

/* compute_site_log_likelihoods_by_class (lib.rs:214-258 -> likelihood.rs:303-338): out[s*K+k] = ln max(f,1e-300) */
int orc_site_log_likelihoods_by_class(int K, int n, const double *Q, const double *pi, int n_nodes,
                                      const int32_t *node_ids, const int32_t *parents, const double *bl,
                                      const int32_t *seq, int n_leaves, int S, const int32_t *leaf_node_ids,
                                      int scaled, int threads, double *out)
{
    orc_tree t;
    int rc = tree_build(&t, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    double *f = (double *)malloc(sizeof(double) * (size_t)K * S);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * S);
    if (!f || !e) { free(f); free(e); tree_free(&t); return ORC_ERR_ALLOC; }
    rc = class_site_likelihoods(&t, K, n, S, Q, pi, bl, seq, scaled, threads, f, e);
    if (rc == ORC_OK)
        for (int s = 0; s < S; s++)
            for (int k = 0; k < K; k++)
                out[(size_t)s * K + k] = ln_clamped(f[(size_t)k * S + s], e[(size_t)k * S + s]);
    free(f); free(e); tree_free(&t);
    return rc;
}

/* compute_log_likelihood_branch (lib.rs:363-430 -> likelihood.rs:385-455): one Q per node.
 * n_q must equal n_nodes (the reference panics otherwise, likelihood.rs:392-398) and
 * len(branch_lengths) (lib.rs:377-384). */
int orc_log_likelihood_branch(int n, int n_q, const double *Q, const double *pi, int n_nodes,
                              const int32_t *parents, const double *bl, int n_bl, const int32_t *seq, int n_leaves,
                              int S, const int32_t *leaf_node_ids, int scaled, int threads, double *lnL, double *site_lnl)
{
    if (n_q != n_bl || n_q != n_nodes) return ORC_ERR_ARG;
    orc_tree t;
    int rc = tree_build(&t, n_nodes, NULL, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    double *P = (double *)malloc(sizeof(double) * (size_t)n_nodes * n * n);
    double *f = (double *)malloc(sizeof(double) * (size_t)S);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
    if (!P || !f || !e) { free(P); free(f); free(e); tree_free(&t); return ORC_ERR_ALLOC; }
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
    for (int v = 0; v < n_nodes; v++) {
        double *U = (double *)malloc(sizeof(double) * (2 * n * n + n));
        if (!U) { rc_all = ORC_ERR_ALLOC; continue; }
        double *Uinv = U + n * n, *lam = Uinv + n * n;
        int r = orc_cached_q(n, Q + (size_t)v * n * n, pi, U, lam, Uinv);   /* lib.rs:387-393 */
        if (r == ORC_OK) orc_expm(n, U, lam, Uinv, bl[v], P + (size_t)v * n * n);  /* likelihood.rs:401-406 */
        else rc_all = r;
        free(U);
    }
    rc = rc_all;
    if (rc == ORC_OK) rc = prune(&t, S, n, P, pi, seq, 0, scaled, f, e);
    if (rc == ORC_OK) {
        double total = 0.0;
        for (int s = 0; s < S; s++) {
            double v = ln_clamped(f[s], e[s]);       /* likelihood.rs:450-451 */
            if (site_lnl) site_lnl[s] = v;
            total += v;
        }
        *lnL = total;
    }
    free(P); free(f); free(e); tree_free(&t);
    return rc;
}

/* compute_branch_site_log_likelihood (lib.rs:287-345 -> branch_site_likelihood.rs:115-299).
 * Q3 = {Q(omega0), Q(1), Q(omega2)} unnormalised; labels[n_labels] with n_labels == n_nodes-1
 * (branch_site_likelihood.rs:41-47), indexed by node id, root (and any id >= n_labels) -> 0 (:61-70).
 * Eigensystem choice follows the float comparisons at :195-201.  lnL = sum_s ln(sum_k freq_k f_ks), no clamp. */
int orc_branch_site_log_likelihood(int n, const double *Q3, double qf_back, double qf_fore, double omega0,
                                   double omega2, const double *freqs, const double *pi, int n_nodes,
                                   const int32_t *node_ids, const int32_t *parents, const double *bl,
                                   const uint8_t *labels, int n_labels, const int32_t *seq, int n_leaves, int S,
                                   const int32_t *leaf_node_ids, int scaled, int threads, double *lnL, double *site_lnl)
{
    if (n_labels != n_nodes - 1) return ORC_ERR_ARG;
    for (int v = 0; v < n_labels; v++) if (labels[v] > 1) return ORC_ERR_ARG;   /* panic at :100 */
    orc_tree t;
    int rc = tree_build(&t, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    double *eig = (double *)malloc(sizeof(double) * 3 * (2 * n * n + n));
    double *f = (double *)malloc(sizeof(double) * 4 * (size_t)S);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)S);
    if (!eig || !f || !e) { free(eig); free(f); free(e); tree_free(&t); return ORC_ERR_ALLOC; }
    size_t esz = 2 * n * n + n;
    for (int q = 0; q < 3 && rc == ORC_OK; q++)
        rc = orc_cached_q(n, Q3 + (size_t)q * n * n, pi, eig + q * esz, eig + q * esz + 2 * n * n, eig + q * esz + n * n);
    if (rc == ORC_OK) {
        int rc_all = ORC_OK;
#ifdef _OPENMP
        if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads > 4 ? 4 : threads)
#endif
        for (int k = 0; k < 4; k++) {
            double *P = (double *)malloc(sizeof(double) * (size_t)n_nodes * n * n);
            if (!P) { rc_all = ORC_ERR_ALLOC; continue; }
            for (int v = 0; v < n_nodes; v++) {
                int lab = v < n_labels ? labels[v] : 0;
                /* select_omega (:85-102) */
                double w = (k == 0) ? omega0 : (k == 1) ? 1.0 : (lab == 0 ? (k == 2 ? omega0 : 1.0) : omega2);
                int q = (w == omega0) ? 0 : (w == 1.0) ? 1 : 2;                  /* :195-201 */
                double qf = lab == 0 ? qf_back : qf_fore;                         /* :204-208 */
                const double *E = eig + q * esz;
                orc_expm(n, E, E + 2 * n * n, E + n * n, bl[v] * qf, P + (size_t)v * n * n);   /* :214 */
            }
            int r = prune(&t, S, n, P, pi, seq, 1, scaled, f + (size_t)k * S, e + (size_t)k * S);
            if (r != ORC_OK) rc_all = r;
            free(P);
        }
        rc = rc_all;
    }
    if (rc == ORC_OK) {
        double total = 0.0;
        for (int s = 0; s < S; s++) {
            double v;
            if (!scaled) {
                double m = 0.0;
                for (int k = 0; k < 4; k++) m += freqs[k] * f[(size_t)k * S + s];   /* :163-166 */
                v = log(m);                                                         /* :167, no clamp */
            } else {
                int32_t emax = INT32_MIN;
                for (int k = 0; k < 4; k++) if (e[(size_t)k * S + s] > emax) emax = e[(size_t)k * S + s];
                double m = 0.0;
                for (int k = 0; k < 4; k++) m += freqs[k] * ldexp(f[(size_t)k * S + s], e[(size_t)k * S + s] - emax);
                v = ln_scaled(m, emax);
            }
            if (site_lnl) site_lnl[s] = v;
            total += v;
        }
        *lnL = total;
    }
    free(eig); free(f); free(e); tree_free(&t);
    return rc;
}

/* Pruning alone, from caller-supplied transition matrices: P[K][n_nodes][n][n] -> out_f[k*S+s] (and the
 * power-of-two exponents out_e when scaled).  naive selects the branch-site calculator's loop.  This is
 * compute_site_likelihoods (likelihood.rs:200-246) minus its expm_multiple call; the parity tests feed it
 * the matrices exported from the device so that kernel 2 is checked independently of kernel 1. */
int orc_site_likelihoods_from_p(int K, int n, const double *P, const double *pi, int n_nodes, const int32_t *node_ids,
                                const int32_t *parents, const int32_t *seq, int n_leaves, int S,
                                const int32_t *leaf_node_ids, int naive, int scaled, int threads, double *out_f,
                                int32_t *out_e)
{
    orc_tree t;
    int rc = tree_build(&t, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
    if (rc != ORC_OK) { tree_free(&t); return rc; }
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int k = 0; k < K; k++) {
        int r = prune(&t, S, n, P + (size_t)k * n_nodes * n * n, pi, seq, naive, scaled, out_f + (size_t)k * S,
                      out_e + (size_t)k * S);
        if (r != ORC_OK) rc_all = r;
    }
    tree_free(&t);
    return rc_all;
}

/* ------------------------------------------------------------------ */
/* build_codon_q_matrix (rust/src/qmatrix.rs:22-80) over the codon graph  */
/* of rust/src/codon.rs:63-87,167-220 (PAML order, T C A G, stops removed) */
/* ------------------------------------------------------------------ */
static const char AA64[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";

int orc_build_codon_q(double kappa, double omega, const double *pi, int has_factor, double factor, double *Q)
{
    int sense[61], idx64[64], m = 0;
    for (int c = 0; c < 64; c++) { idx64[c] = -1; if (AA64[c] != '*') { sense[m] = c; idx64[c] = m++; } }
    memset(Q, 0, sizeof(double) * 61 * 61);
    for (int i = 0; i < 61; i++) {
        int ci = sense[i];
        for (int pos = 0; pos < 3; pos++) {
            int shift = 2 * (2 - pos), ni = (ci >> shift) & 3;
            for (int nj = 0; nj < 4; nj++) {
                if (nj == ni) continue;
                int cj = (ci & ~(3 << shift)) | (nj << shift);
                int j = idx64[cj];
                if (j < 0) continue;
                double rate = 1.0;
                /* T=0 C=1 A=2 G=3: transitions are T<->C and A<->G */
                if ((ni ^ nj) == 1) rate *= kappa;
                if (AA64[ci] != AA64[cj]) rate *= omega;
                Q[i * 61 + j] = rate * pi[j];
            }
        }
    }
    for (int i = 0; i < 61; i++) {
        double rs = 0.0;
        for (int j = 0; j < 61; j++) rs += Q[i * 61 + j];
        Q[i * 61 + i] = -rs;
    }
    double div = factor;
    if (!has_factor) {
        double er = 0.0;
        for (int i = 0; i < 61; i++) er += pi[i] * Q[i * 61 + i];
        er = -er;
        if (!(er > 0.0)) return ORC_OK;   /* qmatrix.rs:73-78 */
        div = er;
    }
    for (int i = 0; i < 61 * 61; i++) Q[i] /= div;
    return ORC_OK;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
