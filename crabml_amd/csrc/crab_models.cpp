// Host side of libcrabml_amd.so: rate matrices and the model-level entry points (see crab_host.h).
#include "crab_host.h"

namespace crab_host {


// ---------------------------------------------------------------------------------------------
// the codon graph: ordered single-nucleotide neighbour pairs among the 61 sense codons in PAML order
// (codon.rs:63-87,167-220), row by row with ascending column
// ---------------------------------------------------------------------------------------------

const CodonGraph &codon_graph()
{
    static const CodonGraph g = [] {
        CodonGraph g{};
        static const char aa[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
        int sense[61], rank[64], m = 0;
        for (int c = 0; c < 64; ++c) {
            rank[c] = -1;
            if (aa[c] != '*') {
                sense[m] = c;
                rank[c] = m++;
            }
        }
        int n = 0;
        for (int i = 0; i < 61; ++i) {
            g.row_begin[i] = n;
            for (int j = 0; j < 61; ++j) {
                const int x = sense[i] ^ sense[j];
                int pos = -1;
                if (x != 0 && (x & ~0x03) == 0) pos = 2;
                else if (x != 0 && (x & ~0x0c) == 0) pos = 1;
                else if (x != 0 && (x & ~0x30) == 0) pos = 0;
                if (pos < 0) continue;
                const int sh = 2 * (2 - pos), ni = (sense[i] >> sh) & 3, nj = (sense[j] >> sh) & 3;
                g.col[n] = j;
                g.ts[n] = (ni ^ nj) == 1;                     // T<->C (0,1) and A<->G (2,3)
                g.nonsyn[n] = aa[sense[i]] != aa[sense[j]];
                ++n;
            }
        }
        g.row_begin[61] = n;
        return g;
    }();
    return g;
}

// mu(kappa, omega) = -sum_i pi_i Q_ii of the unnormalised matrix, with the sums taken in the order
// crab_build_codon_q_matrix takes them (same bits as building the matrix and reading its diagonal)
double codon_mean_rate(double kappa, double omega, const double *pi)
{
    const CodonGraph &g = codon_graph();
    double er = 0.0;
    for (int i = 0; i < 61; ++i) {
        double rs = 0.0;
        for (int k = g.row_begin[i]; k < g.row_begin[i + 1]; ++k) rs += g.rate(k, kappa, omega) * pi[g.col[k]];
        er += pi[i] * -rs;
    }
    return -er;
}


}  // namespace crab_host

extern "C" {

// build_codon_q_matrix (qmatrix.rs:22-80) over the single-nucleotide neighbour graph of the 61 sense
// codons in PAML order (codon.rs:63-87,167-220).  Host arithmetic; 526 non-zero off-diagonals.
int crab_build_codon_q_matrix(double kappa, double omega, const double *pi, int has_factor, double factor, double *Q)
{
    if (!pi || !Q) return CRAB_ERR_ARG;
    const CodonGraph &g = codon_graph();
    std::fill(Q, Q + 61 * 61, 0.0);
    for (int i = 0; i < 61; ++i) {
        double rs = 0.0;
        for (int k = g.row_begin[i]; k < g.row_begin[i + 1]; ++k) {
            const double q = g.rate(k, kappa, omega) * pi[g.col[k]];
            Q[i * 61 + g.col[k]] = q;
            rs += q;
        }
        Q[i * 61 + i] = -rs;
    }
    double div = factor;
    if (!has_factor) {
        double er = 0.0;
        for (int i = 0; i < 61; ++i) er += pi[i] * Q[i * 61 + i];
        er = -er;
        if (!(er > 0.0)) return CRAB_OK;
        div = er;
    }
    for (int i = 0; i < 61 * 61; ++i) Q[i] /= div;
    return CRAB_OK;
}

}  // extern "C"

// ---- Q construction behind the boundary ---------------------------------------------------------
// crabML builds every rate matrix in Python before each FFI call (models/codon.py:166-246 per matrix; the
// site-class builders :350-376, :441-461, :518-536, :607-627, :709-729, :810-830 ...; codon_branch.py:129-161;
// optimize/branch_site.py:150-170).  These entry points take the model parameters themselves: the FFI carries
// K + a few doubles instead of K x 61 x 61, and B parameter vectors travel in one call.
namespace {

double mean_rate61(const double *Q, const double *pi)
{
    double er = 0.0;
    for (int i = 0; i < 61; ++i) er += pi[i] * Q[i * 61 + i];
    return -er;
}

// distinct (kappa, omega) pairs of a call -> index of their matrix
struct QPool {
    std::vector<std::pair<double, double>> keys;
    std::unordered_map<uint64_t, std::vector<int>> by_hash;
    int get(double kappa, double omega)
    {
        const double kw[2] = {kappa, omega};
        auto &c = by_hash[hash_words(kw, sizeof(kw), 3)];
        for (int i : c)
            if (std::memcmp(&keys[i], kw, sizeof(kw)) == 0) return i;
        c.push_back((int)keys.size());
        keys.emplace_back(kappa, omega);
        return (int)keys.size() - 1;
    }
};

void check_params(const char *who, double kappa, const double *omegas, int n)
{
    if (!(kappa > 0.0) || !std::isfinite(kappa)) fail(CRAB_ERR_ARG, std::string(who) + ": kappa must be positive and finite");
    for (int i = 0; i < n; ++i)
        if (!(omegas[i] >= 0.0) || !std::isfinite(omegas[i])) fail(CRAB_ERR_ARG, std::string(who) + ": omega must be >= 0 and finite");
}

// the shared-scale mixture of the site-class models: Q_k = Q(kappa, omega_k, pi; factor 1) / sum_k p_k mu_k
// (e.g. models/codon.py:709-729)
void build_site_class_qs(int K, double kappa, const double *omegas, const double *props, const double *pi, double *Q)
{
    double scale = 0.0;
    for (int k = 0; k < K; ++k) {
        crab_build_codon_q_matrix(kappa, omegas[k], pi, 1, 1.0, Q + (size_t)k * 61 * 61);
        scale += props[k] * mean_rate61(Q + (size_t)k * 61 * 61, pi);
    }
    if (!(scale > 0.0)) fail(CRAB_ERR_ARG, "site-class rate matrices have zero mean rate");
    for (size_t i = 0; i < (size_t)K * 61 * 61; ++i) Q[i] /= scale;
}

// distinct unnormalised matrices Q(kappa_b, omega_bk); the mixture's shared scale sum_k p_k mu_k
// (e.g. models/codon.py:709-729) divides the branch lengths instead of the matrices.  bl_stride = n_nodes for one
// vector of branch lengths per parameter vector, 0 for one shared by all.
void plan_codon_site_classes(crab_engine *e, EvalPlan &pl, int B, int K, const double *kappa, const double *omegas,
                             const double *proportions, const double *pi, const double *branch_lengths, size_t bl_stride)
{
    const int nn = e->n_nodes;
    QPool qp;
    pl.n_groups = B;
    pl.K = B * K;
    pl.M = B * K * nn;
    pl.eig_id.resize(pl.M);
    pl.t_eff.resize(pl.M);
    pl.pidx.resize((size_t)pl.K * nn);
    for (int b = 0; b < B; ++b) {
        double scale = 0.0;
        for (int k = 0; k < K; ++k) scale += proportions[(size_t)b * K + k] * codon_mean_rate(kappa[b], omegas[(size_t)b * K + k], pi);
        if (!(scale > 0.0)) fail(CRAB_ERR_ARG, "site-class rate matrices have zero mean rate");
        for (int k = 0; k < K; ++k) {
            const int q = qp.get(kappa[b], omegas[(size_t)b * K + k]);
            for (int v = 0; v < nn; ++v) {
                const int m = (b * K + k) * nn + v;
                pl.eig_id[m] = q;
                pl.t_eff[m] = v == e->root ? 0.0 : branch_lengths[(size_t)b * bl_stride + v] / scale;
                pl.pidx[m] = m;
            }
        }
    }
    pl.kw = qp.keys;
    pl.n_eig = (int)qp.keys.size();
    pl.props.assign(proportions, proportions + (size_t)B * K);
}

// Model A's class frequencies [p0, p1, p2 p0/(p0+p1), p2 p1/(p0+p1)] (codon_branch_site.py:108-161) and Qfactors
// 1 / mean rate at the branch type's average omega (:163-231)
void model_a_quantities(double kappa, double omega0, double omega2, double p0, double p1, const double *pi, double *freqs, double *qfactors)
{
    const double t01 = p0 + p1;
    if (!(t01 < 1.0)) fail(CRAB_ERR_ARG, "p0 + p1 must be < 1 (got " + std::to_string(t01) + ")");
    if (t01 < 1e-10) fail(CRAB_ERR_ARG, "p0 + p1 too small (" + std::to_string(t01) + ")");
    const double p2 = 1.0 - p0 - p1;
    freqs[0] = p0;
    freqs[1] = p1;
    freqs[2] = p2 * p0 / t01;
    freqs[3] = p2 * p1 / t01;
    const double w_back = (p0 / t01) * omega0 + (p1 / t01) * 1.0;
    const double w_fore = p0 * omega0 + p1 * 1.0 + p2 * omega2;
    qfactors[0] = 1.0 / codon_mean_rate(kappa, w_back, pi);
    qfactors[1] = 1.0 / codon_mean_rate(kappa, w_fore, pi);
}

}  // namespace

extern "C" {

int crab_model_a_quantities(double kappa, double omega0, double omega2, double p0, double p1, const double *pi, double *site_class_freqs,
                            double *qfactors)
{
    return guarded(nullptr, [&] {
        if (!pi || !site_class_freqs || !qfactors) fail(CRAB_ERR_ARG, "crab_model_a_quantities: bad arguments");
        const double w3[3] = {omega0, 1.0, omega2};
        check_params("crab_model_a_quantities", kappa, w3, 3);
        model_a_quantities(kappa, omega0, omega2, p0, p1, pi, site_class_freqs, qfactors);
    });
}

// Median quantiles (2j+1)/(2K), j = 0..K-1: the class omegas of M7 / M8 / M8a (models/codon.py:597-604, 694-701, 795-802;
// kind 0: beta(a, b)) and of M5 (:892-899; kind 1: gamma with shape a and rate b), with the reference's floor of 0.005
// on both shape parameters (:570-571, :864-865).
int crab_discretise_omega(int kind, double a, double b, int K, double *omegas_out)
{
    return guarded(nullptr, [&] {
        if (K < 1 || !omegas_out || (kind != 0 && kind != 1)) fail(CRAB_ERR_ARG, "crab_discretise_omega: bad arguments");
        if (!std::isfinite(a) || !std::isfinite(b)) fail(CRAB_ERR_ARG, "crab_discretise_omega: shape parameters must be finite");
        a = std::max(a, 0.005);
        b = std::max(b, 0.005);
        for (int j = 0; j < K; ++j) {
            const double p = (2.0 * j + 1.0) / (2.0 * K);
            omegas_out[j] = kind == 0 ? beta_ppf(p, a, b) : gamma_ppf(p, a, 1.0 / b);
        }
    });
}

int crab_discretise_omega_mixture(int kind, const double *params, int n_params, int K, double *omegas_out)
{
    return guarded(nullptr, [&] {
        if (K < 1 || !omegas_out || !params || (kind != 2 && kind != 3) || n_params != (kind == 2 ? 4 : 5))
            fail(CRAB_ERR_ARG, "crab_discretise_omega_mixture: kind 2 (M6) takes 4 parameters, kind 3 (M9) takes 5");
        for (int i = 0; i < n_params; ++i)
            if (!std::isfinite(params[i])) fail(CRAB_ERR_ARG, "crab_discretise_omega_mixture: parameters must be finite");
        if (!mixture_median_quantiles(kind, params, K, omegas_out))
            fail(CRAB_ERR_ARG, "crab_discretise_omega_mixture: Brent's iteration did not converge in 100 steps");
    });
}

int crab_build_site_class_q_matrices(int K, double kappa, const double *omegas, const double *proportions, const double *pi,
                                     double *Q_out)
{
    return guarded(nullptr, [&] {
        if (K < 1 || !omegas || !proportions || !pi || !Q_out) fail(CRAB_ERR_ARG, "crab_build_site_class_q_matrices: bad arguments");
        check_params("crab_build_site_class_q_matrices", kappa, omegas, K);
        build_site_class_qs(K, kappa, omegas, proportions, pi, Q_out);
    });
}

int crab_eval_codon_site_classes(crab_engine *e, int B, int K, const double *kappa, const double *omegas, const double *proportions,
                                 const double *pi, const double *branch_lengths, unsigned flags, double *lnL)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, B, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_codon_site_classes(sh, B, K, kappa, omegas, proportions, pi, branch_lengths, flags, part);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (B < 1 || K < 1 || !kappa || !omegas || !proportions || !pi || !branch_lengths || !lnL)
            fail(CRAB_ERR_ARG, "crab_eval_codon_site_classes: bad arguments");
        if (e->n_states != 61) fail(CRAB_ERR_ARG, "codon-model entry points need a 61-state alignment");
        for (int b = 0; b < B; ++b) check_params("crab_eval_codon_site_classes", kappa[b], omegas + (size_t)b * K, K);
        EvalPlan pl;
        plan_codon_site_classes(e, pl, B, K, kappa, omegas, proportions, pi, branch_lengths, e->n_nodes);
        pl.mix_mode = (flags & CRAB_FLAG_NO_CLAMP) ? MIX_LOGSUMEXP_NOCLAMP : MIX_LOGSUMEXP_CLAMP;
        run_plan(e, pl, pi, lnL, nullptr, nullptr);
    });
}

int crab_eval_codon_class_posteriors(crab_engine *e, int B, int K, const double *kappa, const double *omegas, const double *proportions,
                                     const double *weights, const double *pi, const double *branch_lengths, unsigned flags,
                                     double *posteriors)
{
    (void)flags;
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())   // every shard writes the rows of its own sites
        return fan_out(e, [&](crab_engine *sh, int) {
            return crab_eval_codon_class_posteriors(sh, B, K, kappa, omegas, proportions, weights, pi, branch_lengths, flags, posteriors);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (B < 1 || K < 1 || !kappa || !omegas || !proportions || !weights || !pi || !branch_lengths || !posteriors)
            fail(CRAB_ERR_ARG, "crab_eval_codon_class_posteriors: bad arguments");
        if (K > POST_MAX_K) fail(CRAB_ERR_ARG, "crab_eval_codon_class_posteriors: at most 32 classes");
        if (e->n_states != 61) fail(CRAB_ERR_ARG, "codon-model entry points need a 61-state alignment");
        for (int b = 0; b < B; ++b) check_params("crab_eval_codon_class_posteriors", kappa[b], omegas + (size_t)b * K, K);
        EvalPlan pl;
        plan_codon_site_classes(e, pl, B, K, kappa, omegas, proportions, pi, branch_lengths, 0);
        pl.post_weights = weights;
        pl.post_out = posteriors;
        run_plan(e, pl, pi, nullptr, nullptr, nullptr);
    });
}

int crab_eval_codon_branch(crab_engine *e, int B, const double *kappa, const double *omega_per_node, const double *pi,
                           const double *branch_lengths, unsigned flags, double *lnL)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, B, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_codon_branch(sh, B, kappa, omega_per_node, pi, branch_lengths, flags, part);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (B < 1 || !kappa || !omega_per_node || !pi || !branch_lengths || !lnL) fail(CRAB_ERR_ARG, "crab_eval_codon_branch: bad arguments");
        if (e->n_states != 61) fail(CRAB_ERR_ARG, "codon-model entry points need a 61-state alignment");
        const int nn = e->n_nodes;
        for (int b = 0; b < B; ++b) check_params("crab_eval_codon_branch", kappa[b], omega_per_node + (size_t)b * nn, nn);
        // one eigensystem per distinct (kappa, omega) (codon_branch.py:154-161), one class per vector
        QPool pool;
        EvalPlan pl;
        pl.n_groups = B;
        pl.K = B;
        pl.M = B * nn;
        pl.eig_id.resize(pl.M);
        pl.t_eff.resize(pl.M);
        pl.pidx.resize(pl.M);
        for (int b = 0; b < B; ++b)
            for (int v = 0; v < nn; ++v) {
                const int m = b * nn + v;
                pl.eig_id[m] = pool.get(kappa[b], omega_per_node[(size_t)b * nn + v]);
                pl.t_eff[m] = v == e->root ? 0.0 : branch_lengths[(size_t)b * nn + v];
                pl.pidx[m] = m;
            }
        pl.kw = pool.keys;
        pl.n_eig = (int)pool.keys.size();
        std::vector<double> mu(pl.n_eig);   // each matrix is normalised on its own: t / mu
        parallel_for(pl.n_eig, [&](int i) { mu[i] = codon_mean_rate(pool.keys[i].first, pool.keys[i].second, pi); });
        for (int m = 0; m < pl.M; ++m) {
            if (!(mu[pl.eig_id[m]] > 0.0)) fail(CRAB_ERR_ARG, "rate matrix with zero mean rate");
            pl.t_eff[m] /= mu[pl.eig_id[m]];
        }
        pl.props.assign(B, 1.0);
        pl.mix_mode = (flags & CRAB_FLAG_NO_CLAMP) ? MIX_LOGSUMEXP_NOCLAMP : MIX_LOGSUMEXP_CLAMP;
        run_plan(e, pl, pi, lnL, nullptr, nullptr);
    });
}

int crab_eval_codon_branch_site(crab_engine *e, int B, const double *kappa, const double *omega0, const double *omega2, const double *p0,
                                const double *p1, const double *pi, const double *branch_lengths, const uint8_t *labels, int n_labels,
                                unsigned flags, double *lnL)
{
    (void)flags;
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, B, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_codon_branch_site(sh, B, kappa, omega0, omega2, p0, p1, pi, branch_lengths, labels, n_labels, flags, part);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (B < 1 || !kappa || !omega0 || !omega2 || !p0 || !p1 || !pi || !branch_lengths || !labels || !lnL)
            fail(CRAB_ERR_ARG, "crab_eval_codon_branch_site: bad arguments");
        if (e->n_states != 61) fail(CRAB_ERR_ARG, "codon-model entry points need a 61-state alignment");
        const int nn = e->n_nodes;
        if (n_labels != nn - 1)
            fail(CRAB_ERR_ARG, "Branch labels length " + std::to_string(n_labels) + " must equal n_nodes - 1 = " + std::to_string(nn - 1));
        for (int v = 0; v < n_labels; ++v)
            if (labels[v] > 1) fail(CRAB_ERR_ARG, "Invalid branch_label or site_class");
        QPool pool;
        EvalPlan pl;
        pl.n_groups = B;
        pl.K = 4 * B;
        pl.pidx.assign((size_t)pl.K * nn, 0);
        pl.props.resize(pl.K);
        pl.mix_mode = MIX_LINEAR;
        for (int b = 0; b < B; ++b) {
            const double w3[3] = {omega0[b], 1.0, omega2[b]};
            check_params("crab_eval_codon_branch_site", kappa[b], w3, 3);
            double *fr = &pl.props[(size_t)4 * b];
            double qf[2];
            model_a_quantities(kappa[b], omega0[b], omega2[b], p0[b], p1[b], pi, fr, qf);
            const double qf_back = qf[0], qf_fore = qf[1];
            // eigensystem per (label, class): select_omega (branch_site_likelihood.rs:85-102) and the float
            // comparisons of :195-201, as crab_eval_branch_site
            auto pick = [&](int lab, int k) {
                const double w = k == 0 ? omega0[b] : k == 1 ? 1.0 : (lab == 0 ? (k == 2 ? omega0[b] : 1.0) : omega2[b]);
                return w == omega0[b] ? 0 : (w == 1.0 ? 1 : 2);
            };
            int eig_of[3];
            for (int q = 0; q < 3; ++q) eig_of[q] = pool.get(kappa[b], w3[q]);
            for (int v = 0; v < nn; ++v) {
                const int lab = v < n_labels ? labels[v] : 0;
                const double t = (v == e->root ? 0.0 : branch_lengths[(size_t)b * nn + v]) * (lab == 0 ? qf_back : qf_fore);
                int slot_of_eig[3] = {-1, -1, -1};
                for (int k = 0; k < 4; ++k) {
                    const int q = pick(lab, k);
                    if (slot_of_eig[q] < 0) {
                        slot_of_eig[q] = pl.M++;
                        pl.eig_id.push_back(eig_of[q]);
                        pl.t_eff.push_back(t);
                    }
                    pl.pidx[((size_t)4 * b + k) * nn + v] = slot_of_eig[q];
                }
            }
        }
        pl.kw = pool.keys;   // unnormalised (optimize/branch_site.py:163-165)
        pl.n_eig = (int)pool.keys.size();
        run_plan(e, pl, pi, lnL, nullptr, nullptr);
    });
}

}  // extern "C"

