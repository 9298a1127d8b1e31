// Host side of libcrabml_amd.so: the C ABI declared in include/crabml_amd.h.
//
// What lives here is what rust/src/lib.rs + the orchestration parts of rust/src/likelihood.rs and
// rust/src/branch_site_likelihood.rs do on the host in the reference: argument validation, tree
// construction, choosing an eigensystem and an effective time for every (branch, class), and the
// final reduction -- re-thought for a device that keeps the tree program and the alignment resident:
//   * the tree is compiled once into a flat post-order "program" (TreeOp list) with Sethi-Ullman child
//     ordering so the register/stack footprint of the on-chip tree walk is minimal;
//   * the alignment is compressed into weighted site patterns once and stored as one byte per code;
//   * per evaluation only the packed eigensystems (K x 66 KB) and a few small tables cross PCIe.
#include "crab_host.h"

namespace crab_host {
thread_local std::string g_last_error;
}

namespace crab {
// for entry points that live in other source files (crab_noalloc.hip): the message crab_last_error(NULL) returns
void set_global_error(const std::string &msg) { g_last_error = msg; }
}  // namespace crab

namespace crab_host {

// CRAB_NO_PATH_UPDATES=1: whole walks for every vector of a batch (A/B timing, and the reference for the bit-equality tests)
bool path_updates_disabled()
{
    const char *sv = std::getenv("CRAB_NO_PATH_UPDATES");
    return sv && sv[0] && sv[0] != '0';
}
// CRAB_NO_TREE_PARTS=1: never cut the tree of a single evaluation into parts (A/B timing)
bool tree_parts_disabled()
{
    const char *sv = std::getenv("CRAB_NO_TREE_PARTS");
    return sv && sv[0] && sv[0] != '0';
}
// CRAB_STATIC_ITEMS=1: the items of a launch are dealt round-robin to its workgroups, as in rounds 1-2 (read once)
bool dynamic_items_enabled()
{
    static const bool on = [] {
        const char *sv = std::getenv("CRAB_STATIC_ITEMS");
        return !(sv && sv[0] && sv[0] != '0');
    }();
    return on;
}
// CRAB_CLADE_LEVEL_LAUNCHES=1: one launch per level of clade nesting, as in round 2 (read when an engine plans its programs)
bool clade_level_launches_forced()
{
    const char *sv = std::getenv("CRAB_CLADE_LEVEL_LAUNCHES");
    return sv && sv[0] && sv[0] != '0';
}
bool clade_tables_disabled()
{
    const char *sv = std::getenv("CRAB_NO_CLADE_TABLES");
    return sv && sv[0] && sv[0] != '0';
}
// Largest save area the partial-update path may allocate (CRAB_PATH_SAVE_GB, default 96 of the 288 GB)
size_t path_save_limit_bytes()
{
    double gb = 96.0;
    if (const char *sv = std::getenv("CRAB_PATH_SAVE_GB")) gb = std::atof(sv);
    return gb > 0.0 ? (size_t)(gb * 1073741824.0) : 0;
}

void check_ready(crab_engine *e)
{
    if (!e->have_tree) fail(CRAB_ERR_STATE, "crab_set_tree has not been called");
    if (!e->have_aln) fail(CRAB_ERR_STATE, "crab_set_alignment has not been called");
    if (e->n_leaf_rows > e->aln_rows)
        fail(CRAB_ERR_ARG, "tree has " + std::to_string(e->n_leaf_rows) + " leaves but the alignment has " +
                               std::to_string(e->aln_rows) + " rows");
}

void run_plan(crab_engine *e, const EvalPlan &pl, const double *pi, double *lnL, double *site_lnl, double *by_class_out)
{
    const int n = e->n_states;
    HIP_CHECK(hipSetDevice(e->device));

    // ---- host: eigensystems.  Identical rate matrices are decomposed once (within the call and across calls: the pool
    // lives on the device), the rest in parallel, one task per matrix (replaces the serial loop at lib.rs:163-169) ----
    const double t0 = now_ms();
    const double *Q = pl.Q;
    const bool by_params = !pl.kw.empty();
    EigPool &pool = e->eig_pool;
    ++pool.clock;
    std::vector<int32_t> uniq_of(pl.n_eig);       // rate matrix -> slot among this call's distinct eigensystems
    std::vector<int> uniq_src;                    // slot -> first rate matrix showing it
    std::vector<uint64_t> qhash(pl.n_eig), qhash2(pl.n_eig, 0);
    const uint64_t pih = hash_words(pi, sizeof(double) * n, 11);
    auto digest = [&](int i) {
        uint64_t d[2];
        hash_words2(Q + (size_t)i * n * n, sizeof(double) * n * n, pih, d);
        qhash[i] = d[0], qhash2[i] = d[1];
    };
    if (by_params) {                              // the caller lists distinct (kappa, omega) pairs
        for (int i = 0; i < pl.n_eig; ++i) {
            const double kw[2] = {pl.kw[i].first, pl.kw[i].second};
            qhash[i] = hash_words(kw, sizeof(kw), pih ^ 0x9e3779b97f4a7c15ull);
            uniq_of[i] = i;
            uniq_src.push_back(i);
        }
    } else {
        if ((size_t)pl.n_eig * n * n * sizeof(double) < (1u << 20))   // a few matrices: cheaper than waking the pool (measured)
            for (int i = 0; i < pl.n_eig; ++i) digest(i);
        else
            parallel_for(pl.n_eig, digest);
        // group by digest (a batch of B vectors repeats almost all of its B x K matrices); with CRAB_EIG_CONFIRM every presumed
        // duplicate is also compared with its representative byte for byte, in parallel, and a collision falls back to the
        // exact search
        auto group = [&](bool exact) {
            uniq_src.clear();
            std::unordered_map<uint64_t, std::vector<int>> seen;
            for (int i = 0; i < pl.n_eig; ++i) {
                const double *Qi = Q + (size_t)i * n * n;
                auto &cands = seen[qhash[i]];
                int slot = -1;
                for (int u : cands)
                    if (qhash2[uniq_src[u]] == qhash2[i] &&
                        (!exact || std::memcmp(Q + (size_t)uniq_src[u] * n * n, Qi, sizeof(double) * n * n) == 0)) {
                        slot = u;
                        break;
                    }
                if (slot < 0) {
                    slot = (int)uniq_src.size();
                    uniq_src.push_back(i);
                    cands.push_back(slot);
                }
                uniq_of[i] = slot;
            }
        };
        group(false);
        const int dupes = pl.n_eig - (int)uniq_src.size();
        if (dupes > 0 && EigPool::confirm_hits()) {
            std::atomic<int> collisions{0};
            auto check = [&](int i) {
                const int rep = uniq_src[uniq_of[i]];
                if (rep != i && std::memcmp(Q + (size_t)rep * n * n, Q + (size_t)i * n * n, sizeof(double) * n * n) != 0) ++collisions;
            };
            if (dupes < 64)
                for (int i = 0; i < pl.n_eig; ++i) check(i);
            else
                parallel_for(pl.n_eig, check);
            if (collisions.load() != 0) group(true);
        }
    }
    const int n_uniq = (int)uniq_src.size();
    const double t_grouped = now_ms();
    std::vector<int32_t> pool_slot(n_uniq, -1);
    std::vector<int> todo;
    auto look_up = [&](int u) {
        const int i = uniq_src[u];
        pool_slot[u] = by_params ? pool.find_params(qhash[i], pl.kw[i].first, pl.kw[i].second, pi, n)
                                 : pool.find_content(qhash[i], qhash2[i], Q + (size_t)i * n * n, pi, n);
    };
    // (With CRAB_EIG_CONFIRM a hit is compared byte for byte with the cached matrix -- 59 MB for the 1 998 matrices of a
    // free-ratio call -- and the look-ups run side by side: they only read the index, distinct matrices stamp distinct entries.)
    if (!by_params && EigPool::confirm_hits() && (size_t)n_uniq * n * n * sizeof(double) >= (1u << 20))
        parallel_for(n_uniq, look_up);
    else
        for (int u = 0; u < n_uniq; ++u) look_up(u);
    for (int u = 0; u < n_uniq; ++u)
        if (pool_slot[u] < 0) todo.push_back(u);
    if (std::getenv("CRAB_DEBUG_EIGEN_TIMES") && n_uniq >= 64)
        fprintf(stderr, "[crab eigen] %d matrices: hashing and grouping %.3f ms, look-ups %.3f ms, %zu to decompose\n", n_uniq, t_grouped - t0,
                now_ms() - t_grouped, todo.size());
    e->eig_hits += n_uniq - (int)todo.size();
    e->eig_misses += (int)todo.size();
    const bool device_eigen = e->device_eigen && !todo.empty();
    e->h_eig.ensure(device_eigen ? 1 : todo.size() * EIG_STRIDE);
    double *h_eig = e->h_eig.p;
    std::vector<char> ok(todo.size(), 1);
    std::vector<double> Qbuilt(by_params ? todo.size() * (size_t)n * n : 0);
    static const bool debug_times = std::getenv("CRAB_DEBUG_EIGEN_TIMES") != nullptr;   // per-task start / end, to stderr
    std::vector<double> dbg_t0(debug_times ? todo.size() : 0), dbg_t1(dbg_t0.size());
    const auto dbg_origin = std::chrono::steady_clock::now();
    auto dbg_now = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_origin).count(); };
    auto decompose_one = [&](int j) {
        struct Stamp {
            double *t0, *t1;
            const decltype(dbg_now) &now;
            Stamp(double *a, double *b, const decltype(dbg_now) &n) : t0(a), t1(b), now(n) { if (t0) *t0 = now(); }
            ~Stamp() { if (t1) *t1 = now(); }
        } stamp(debug_times ? &dbg_t0[j] : nullptr, debug_times ? &dbg_t1[j] : nullptr, dbg_now);
        const int i = uniq_src[todo[j]];
        const double *Qi = Q + (size_t)i * n * n;
        if (by_params) {
            double *dst = Qbuilt.data() + (size_t)j * n * n;
            crab_build_codon_q_matrix(pl.kw[i].first, pl.kw[i].second, pi, 1, 1.0, dst);
            Qi = dst;
        }
        ok[j] = spectral_decompose(n, Qi, pi, h_eig + (size_t)j * EIG_STRIDE) ? 1 : 0;
    };
    if (device_eigen) {
        // The device decomposes (kernel 0): the host only symmetrises, S = Pi^1/2 Q Pi^-1/2 from the lower triangle like
        // eigh(UPLO::Lower) (matrix.rs:42-54), padded to 64 x 64, and ships it with the slot list and sqrt(pi).
        const size_t per = (size_t)NS * NS, n_todo = todo.size();
        e->h_S.ensure(n_todo * per + NS + (n_todo + 1) / 2 + 1);
        double *hS = e->h_S.p;
        std::vector<double> sq(n), isq(n);
        for (int i = 0; i < n; ++i) {
            if (!(pi[i] > 0.0)) fail(CRAB_ERR_EIGEN, "Eigendecomposition failed: pi has a non-positive entry");
            sq[i] = std::sqrt(pi[i]);
            isq[i] = 1.0 / sq[i];
        }
        parallel_for((int)n_todo, [&](int j) {
            const int i0 = uniq_src[todo[j]];
            const double *Qi = Q + (size_t)i0 * n * n;
            std::vector<double> built;
            if (by_params) {
                built.resize((size_t)n * n);
                crab_build_codon_q_matrix(pl.kw[i0].first, pl.kw[i0].second, pi, 1, 1.0, built.data());
                Qi = built.data();
            }
            double *Sj = hS + (size_t)j * per;
            std::memset(Sj, 0, per * sizeof(double));
            bool finite = true;
            for (int a = 0; a < n; ++a)
                for (int b = 0; b <= a; ++b) {
                    const double v = sq[a] * Qi[(size_t)a * n + b] * isq[b];
                    finite = finite && std::isfinite(v);
                    Sj[(size_t)a * NS + b] = v;
                    Sj[(size_t)b * NS + a] = v;
                }
            ok[j] = finite ? 1 : 0;
        });
    } else if (e->shared_eig) {   // a shard of a multi-device handle: every shard needs this matrix, one of them decomposes it
        SharedEigCache *sc = e->shared_eig;
        parallel_for((int)todo.size(), [&](int j) {
            const int i = uniq_src[todo[j]];
            ok[j] = sc->get(qhash[i], by_params, by_params ? pl.kw[i].first : 0.0, by_params ? pl.kw[i].second : 0.0,
                            by_params ? nullptr : Q + (size_t)i * n * n, pi, n, h_eig + (size_t)j * EIG_STRIDE, [&](double *) {
                                decompose_one(j);
                                return ok[j] != 0;
                            })
                        ? 1
                        : 0;
        });
    } else {
        parallel_for((int)todo.size(), decompose_one);
    }
    if (debug_times && !device_eigen && !todo.empty()) {
        std::string line = "[crab eigen] " + std::to_string(todo.size()) + " decompositions, all done at " + std::to_string((int)dbg_now()) + " us:";
        for (size_t j = 0; j < todo.size(); ++j) line += " " + std::to_string((int)dbg_t0[j]) + "-" + std::to_string((int)dbg_t1[j]);
        fprintf(stderr, "%s\n", line.c_str());
    }
    for (size_t j = 0; j < todo.size(); ++j)
        if (!ok[j]) fail(CRAB_ERR_EIGEN, "Eigendecomposition failed for rate matrix " + std::to_string(uniq_src[todo[j]]));
    std::vector<int> new_slots;
    struct SlotGuard {   // slots taken but not registered (an error further down) go back to the free list
        EigPool &pool;
        std::vector<int> &got;
        ~SlotGuard() { pool.give_back(got); }
    } slot_guard{pool, new_slots};
    const double *d_eig_S = nullptr, *d_eig_sqrt_pi = nullptr;
    const int32_t *d_eig_slots = nullptr;
    if (device_eigen) {
        pool.allocate((int)todo.size(), new_slots, e->stream);
        const size_t per = (size_t)NS * NS, n_todo = todo.size(), n_d = n_todo * per + NS + (n_todo + 1) / 2 + 1;
        double *hS = e->h_S.p, *h_sq = hS + n_todo * per;
        for (int i = 0; i < NS; ++i) h_sq[i] = i < n ? std::sqrt(pi[i]) : 0.0;
        int32_t *h_slots = reinterpret_cast<int32_t *>(h_sq + NS);
        for (size_t j = 0; j < n_todo; ++j) h_slots[j] = new_slots[j];
        e->d_S.ensure(n_d);
        HIP_CHECK(hipMemcpyAsync(e->d_S.p, hS, n_d * sizeof(double), hipMemcpyHostToDevice, e->stream));
        d_eig_S = e->d_S.p;
        d_eig_sqrt_pi = e->d_S.p + n_todo * per;
        d_eig_slots = reinterpret_cast<const int32_t *>(e->d_S.p + n_todo * per + NS);
        for (size_t j = 0; j < todo.size(); ++j) pool_slot[todo[j]] = new_slots[j];
    } else if (!todo.empty()) {
        pool.allocate((int)todo.size(), new_slots, e->stream);
        // consecutive slots travel in one copy (a first call fills a fresh pool front to back)
        for (size_t j = 0; j < todo.size();) {
            size_t r = j + 1;
            while (r < todo.size() && new_slots[r] == new_slots[r - 1] + 1) ++r;
            const hipError_t err = hipMemcpyAsync(pool.d + (size_t)new_slots[j] * EIG_STRIDE, h_eig + j * EIG_STRIDE,
                                                  (r - j) * EIG_STRIDE * sizeof(double), hipMemcpyHostToDevice, e->stream);
            if (err != hipSuccess) throw HipError{std::string("uploading eigensystems: ") + hipGetErrorString(err)};
            j = r;
        }
        for (size_t j = 0; j < todo.size(); ++j) pool_slot[todo[j]] = new_slots[j];
    }
    // (the new entries are registered further down, while the device is busy)
    e->last_eigen_ms = now_ms() - t0;

    // ---- transition matrices to build.  In a batch most of them repeat: a finite-difference stencil changes one branch
    // length or one class per vector, so (eigensystem, t) pairs are built once and shared (at C2 a 203-vector gradient
    // would otherwise build 28 GB of matrices; at C3 more than the card holds).  Single evaluations skip the search. ----
    int M = pl.M;
    std::vector<int32_t> mat_slot(M);      // eigensystem pool slot of matrix m
    for (int m = 0; m < M; ++m) mat_slot[m] = pool_slot[uniq_of[pl.eig_id[m]]];
    const std::vector<double> *t_eff = &pl.t_eff;
    const std::vector<int32_t> *pidx = &pl.pidx;
    std::vector<double> t_u;
    std::vector<int32_t> slot_u, pidx_u;
    if (pl.n_groups > 1 && M > 1) {
        size_t cap = 64;
        while (cap < (size_t)M * 2) cap <<= 1;
        std::vector<int32_t> table(cap, -1), remap(M);
        for (int m = 0; m < M; ++m) {
            uint64_t tb;
            std::memcpy(&tb, &pl.t_eff[m], 8);
            size_t h = (size_t)mix64(tb, (uint64_t)mat_slot[m]) & (cap - 1);
            for (;;) {
                const int32_t u = table[h];
                if (u < 0) {
                    table[h] = (int32_t)t_u.size();
                    remap[m] = (int32_t)t_u.size();
                    t_u.push_back(pl.t_eff[m]);
                    slot_u.push_back(mat_slot[m]);
                    break;
                }
                if (slot_u[u] == mat_slot[m] && std::memcmp(&t_u[u], &pl.t_eff[m], 8) == 0) {
                    remap[m] = u;
                    break;
                }
                h = (h + 1) & (cap - 1);
            }
        }
        pidx_u.resize(pl.pidx.size());
        for (size_t i = 0; i < pl.pidx.size(); ++i) pidx_u[i] = remap[pl.pidx[i]];
        M = (int)t_u.size();
        mat_slot.swap(slot_u);
        t_eff = &t_u;
        pidx = &pidx_u;
    }

    const WorkSplit ws = plan_device_work(e, pl, pidx, M);
    const int G = ws.G, Kg = ws.Kg, n_path = ws.n_path, n_parts = ws.n_parts, n_cl = ws.n_cl, n_tab = ws.n_tab;
    const int64_t n_tiles = ws.n_tiles;
    const bool use_clades = ws.use_clades;
    const int n_first = e->dev_n_first, n_second = e->dev_n_second;
    const bool two_launches = n_second > 0;
    // the small per-call inputs travel in one copy: [t_eff M][pi NS][props K][eigensystem slot per matrix, int32 M]
    const size_t n_small = (size_t)M + NS + pl.K + ((size_t)M + 1) / 2;
    e->h_small.ensure(n_small);
    double *hs = e->h_small.p;
    std::memcpy(hs, t_eff->data(), M * sizeof(double));
    for (int i = 0; i < NS; ++i) hs[M + i] = i < n ? pi[i] : 0.0;
    for (int k = 0; k < pl.K; ++k) hs[M + NS + k] = pl.props.empty() ? 1.0 : pl.props[k];
    int32_t *h_ids = reinterpret_cast<int32_t *>(hs + M + NS + pl.K);
    {   // which image of each matrix has a reader: leaf edges are gathered from, all other edges multiplied with
        std::vector<uint8_t> used(M, 0);
        const int nn = e->n_nodes;
        const size_t n_cls = pidx->size() / (size_t)nn;
        for (size_t c = 0; c < n_cls; ++c)
            for (int v = 0; v < nn; ++v)
                if (v != e->root && e->reachable[v]) used[(*pidx)[c * nn + v]] |= e->in_leaf_set[v] ? 1 : 2;
        for (int m = 0; m < M; ++m)   // (a matrix nobody reads keeps both images: crab_export_transition_matrices shows it)
            h_ids[m] = mat_slot[m] | (used[m] == 1 ? BUILD_P_NO_OPERAND : used[m] == 2 ? BUILD_P_NO_GATHER : 0);
    }
    if (M & 1) h_ids[M] = 0;

    e->d_small.ensure(n_small);
    double *d_teff = e->d_small.p, *d_pi = d_teff + M, *d_props = d_pi + NS;
    int32_t *d_eig_id = reinterpret_cast<int32_t *>(d_props + pl.K);
    e->d_P.ensure(((size_t)M + (use_clades ? (size_t)n_tab * e->clades.slots_per_class : 0)) * P_ELEMS);
    int64_t n_items_max = n_tiles * n_first;
    if (use_clades)
        for (const auto &cl : e->dev_clade_launches) n_items_max = std::max(n_items_max, cl.n_items);
    const int grid = prune_grid_size(n_tiles * n_first, e->device);
    const int grid_second = two_launches ? prune_grid_size(n_tiles * n_second, e->device) : 0;
    const int grid_max = std::max({grid, grid_second, prune_grid_size(n_items_max, e->device)});
    e->d_stack.ensure(prune_stack_doubles(grid_max, e->max_depth));
    e->d_stack_e.ensure(prune_stack_ints(grid_max, e->max_depth));
    e->d_f.ensure((size_t)pl.K * e->n_pat_pad);
    e->d_e.ensure((size_t)pl.K * e->n_pat_pad);
    e->d_patlnl.ensure((size_t)G * e->n_pat_pad);
    e->d_partials.ensure((size_t)G * std::max(1, mix_partials(e->n_pat)));
    e->d_lnL.ensure(G);
    if (pl.by_class) e->d_byclass.ensure((size_t)e->n_pat * pl.K + 1);

    HIP_CHECK(hipMemcpyAsync(e->d_small.p, hs, n_small * sizeof(double), hipMemcpyHostToDevice, e->stream));

    // ---- device ----
    HIP_CHECK(hipEventRecord(e->ev_begin, e->stream));
    if (device_eigen) {
        e->d_eig_status.ensure(1);
        e->h_eig_status.ensure(1);
        HIP_CHECK(hipMemsetAsync(e->d_eig_status.p, 0, sizeof(int32_t), e->stream));
        launch_eigen(d_eig_S, d_eig_slots, d_eig_sqrt_pi, pool.d, (int)todo.size(), n, e->d_eig_status.p, e->stream);
        HIP_CHECK(hipMemcpyAsync(e->h_eig_status.p, e->d_eig_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    }
    // Every launch of the pruning kernel hands its items out through eight counters of its own (PruneArgs::work_ctr;
    // CRAB_STATIC_ITEMS=1: items dealt round-robin instead, as in rounds 1-2).  Behind them in the same buffer: the fault flag,
    // the per-XCD totals and the per-program table counters of a merged clade launch.  All zero before the first pruning launch:
    // kernel 1, which runs ahead of them on the stream, clears the buffer (a memset of 3 KB was two fill launches per step).
    constexpr int MAX_PRUNE_LAUNCHES = 16;
    constexpr int SYNC_AT = 8 * MAX_PRUNE_LAUNCHES;   // (PruneArgs::sync: the flag, the totals, the counters)
    const bool dynamic_items = dynamic_items_enabled();
    const bool merged = use_clades && !e->dev_clade_launches.empty() && e->dev_clade_launches[0].merged;
    if (merged && !dynamic_items) fail(CRAB_ERR_ARG, "merged clade launch without dynamic work items");
    size_t n_ctr = 0;
    if (dynamic_items) {
        n_ctr = (size_t)SYNC_AT + 16 + (merged ? (size_t)e->dev_n_prog : 0);
        e->d_work_ctr.ensure(n_ctr);
        if ((int)e->dev_clade_launches.size() + 2 > MAX_PRUNE_LAUNCHES) fail(CRAB_ERR_ARG, "too many clade levels");
    }
    launch_build_p(pool.d, d_eig_id, d_teff, e->d_P.p, M, n, true, e->stream, n_ctr ? e->d_work_ctr.p : nullptr, (int)n_ctr);

    PruneArgs pa{};
    pa.ops = e->d_ops.p;
    pa.n_ops = (int32_t)e->ops.size();
    pa.n_ops_pad = (int32_t)dev_ops_padded((int64_t)e->ops.size());
    pa.P = e->d_P.p;
    pa.K = n_first;
    const bool lists_used = two_launches || use_clades;
    pa.hdr = lists_used ? e->d_hdr.p : nullptr;
    pa.cls_list = lists_used ? e->d_cls.p + e->dev_first_off : nullptr;
    pa.save = (n_path > 0 || n_parts > 0) ? (char *)e->d_save.p : nullptr;
    pa.save_K = Kg;
    std::copy(e->dev_bounds_first, e->dev_bounds_first + 9, pa.xcd_bounds);
    pa.codes = e->d_codes.p;
    pa.n_pat_pad = e->n_pat_pad;
    pa.pi_pad = d_pi;
    pa.n_states = n;
    pa.stack = e->d_stack.p;
    pa.stack_e = e->d_stack_e.p;
    pa.max_depth = std::max(1, e->max_depth);
    pa.out_f = e->d_f.p;
    pa.out_e = e->d_e.p;
    pa.first_gemm_op = e->first_gemm_op;
    pa.first_leaf_row = e->first_leaf_row;
    pa.first_leaf_op = e->first_leaf_op;
    pa.second_leaf_row = e->second_leaf_row;
    pa.third_leaf_row = e->third_leaf_row;
    pa.second_leaf_op = e->second_leaf_op;
    pa.fourth_leaf_row = e->fourth_leaf_row;
    int launch_no = 0;
    HIP_CHECK(hipEventRecord(e->ev_k0, e->stream));
    if (use_clades) {     // one launch per level: the clade programs, over each clade's distinct sub-patterns
        PruneArgs pc = pa;
        pc.save = nullptr;
        pc.n_code_rows = e->aln_rows;
        for (size_t l = 0; l < e->dev_clade_launches.size(); ++l) {
            const crab_engine::CladeLaunch &cl = e->dev_clade_launches[l];
            pc.K = cl.n_entries;
            pc.cls_list = e->d_cls.p + cl.list_off;
            pc.items = e->d_items.p + cl.items_off;
            pc.n_items = cl.n_items;
            pc.codes_alt = e->d_codes_alt.p;   // (a program's header names its level's slice; wide rows are numbered over all launches)
            pc.idx32 = e->d_idx32.p;
            if (cl.merged) {
                pc.deps = e->d_deps.p;
                pc.sync = e->d_work_ctr.p + SYNC_AT;
            }
            std::copy(cl.bounds, cl.bounds + 9, pc.xcd_bounds);
            pc.work_ctr = dynamic_items ? e->d_work_ctr.p + 8 * launch_no++ : nullptr;
            launch_prune(pc, cl.grid, e->stream);
        }
        // the main programs (among the first launch's) read the outermost clades' tables through the last block of wide rows
        pa.idx32 = e->d_idx32.p;
        pa.n_code_rows = e->aln_rows;
        if (merged) {   // an XCD that ran no workgroup, a workgroup that gave up waiting: NaN instead of stale tables
            pa.sync = e->d_work_ctr.p + SYNC_AT;
            pa.expect_tiles = e->dev_clade_tiles;
            e->h_sync_flag.ensure(1);
            e->h_sync_flag.p[0] = 0;
            pa.host_flag = e->h_sync_flag.dev;
            if (const char *sv = std::getenv("CRAB_DEBUG_SYNC"))   // "2": pretend a tile went missing (the tests' way to see the NaN)
                if (sv[0] == '2') pa.expect_tiles += 1;
        }
    }
    pa.work_ctr = dynamic_items ? e->d_work_ctr.p + 8 * launch_no++ : nullptr;
    launch_prune(pa, grid, e->stream);
    if (two_launches) {   // path programs / root programs read what the first launch saved: same stream, after it
        pa.K = n_second;
        pa.cls_list = e->d_cls.p + e->dev_first_off + n_first;
        std::copy(e->dev_bounds_second, e->dev_bounds_second + 9, pa.xcd_bounds);
            pa.work_ctr = dynamic_items ? e->d_work_ctr.p + 8 * launch_no++ : nullptr;
        launch_prune(pa, grid_second, e->stream);
    }
    HIP_CHECK(hipEventRecord(e->ev_k1, e->stream));
    if (merged && std::getenv("CRAB_DEBUG_SYNC")) {
        int32_t h[16];
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(h, e->d_work_ctr.p + SYNC_AT, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[crab sync] flag %d: %d workgroup(s) gave up waiting; one of them: block %d on XCC %d\n", h[0], h[1], h[2] >> 8, h[2] & 255);
    }

    e->h_out.ensure(G + (site_lnl ? (size_t)e->n_pat_pad : 0) + (pl.by_class ? (size_t)e->n_pat * pl.K : 0) +
                    (pl.beb ? (size_t)2 * pl.beb->G + (size_t)e->n_pat * pl.beb->n_mix : 0) +
                    (pl.post_weights ? (size_t)e->n_pat * (pl.K / G) : 0));
    if (pl.post_weights) {
        const int Kc = pl.K / G;
        e->d_beb_w.ensure(G);
        e->d_beb_post.ensure((size_t)std::max<int64_t>(1, e->n_pat) * Kc);
        HIP_CHECK(hipMemcpyAsync(e->d_beb_w.p, pl.post_weights, G * sizeof(double), hipMemcpyHostToDevice, e->stream));
        launch_class_posteriors(e->d_f.p, e->d_e.p, G, Kc, e->n_pat, e->n_pat_pad, d_props, e->d_beb_w.p, e->d_beb_post.p, e->stream);
        if (e->n_pat > 0)
            HIP_CHECK(hipMemcpyAsync(e->h_out.p + G, e->d_beb_post.p, (size_t)e->n_pat * Kc * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    } else if (pl.beb) {
        const EvalPlan::Beb &bb = *pl.beb;
        const size_t gm = (size_t)bb.G * bb.n_mix;
        e->d_beb_F.ensure((size_t)pl.K * e->n_pat_pad);
        e->d_beb_cols.ensure(gm);
        e->d_beb_props.ensure(gm);
        e->d_beb_lnfx.ensure(bb.G);
        e->d_beb_w.ensure(bb.G);
        e->d_beb_post.ensure((size_t)std::max<int64_t>(1, e->n_pat) * bb.n_mix);
        // (the grid tables are caller memory, borrowed until this call returns)
        HIP_CHECK(hipMemcpyAsync(e->d_beb_cols.p, bb.cols, gm * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        HIP_CHECK(hipMemcpyAsync(e->d_beb_props.p, bb.props, gm * sizeof(double), hipMemcpyHostToDevice, e->stream));
        BebArgs ba{};
        ba.f = e->d_f.p;
        ba.e = e->d_e.p;
        ba.K = pl.K;
        ba.n_pat = e->n_pat;
        ba.n_pat_pad = e->n_pat_pad;
        ba.weights = e->d_weights.p;
        ba.F = e->d_beb_F.p;
        ba.G = bb.G;
        ba.n_mix = bb.n_mix;
        ba.cols = e->d_beb_cols.p;
        ba.props = e->d_beb_props.p;
        ba.lnfx = e->d_beb_lnfx.p;
        ba.grid_w = e->d_beb_w.p;
        ba.post = e->d_beb_post.p;
        double *h_lnfx = e->h_out.p + G, *h_w = h_lnfx + bb.G, *h_post = h_w + bb.G;
        launch_beb_scale(ba, e->stream);
        if (bb.mode != EvalPlan::Beb::GIVEN) {
            launch_beb_grid_lnf(ba, e->stream);
            HIP_CHECK(hipMemcpyAsync(h_lnfx, e->d_beb_lnfx.p, bb.G * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            HIP_CHECK(hipStreamSynchronize(e->stream));
            if (e->n_pat == 0)
                for (int g = 0; g < bb.G; ++g) h_lnfx[g] = 0.0;
        } else {
            std::copy(bb.lnfx, bb.lnfx + bb.G, h_lnfx);   // summed over all pattern slices by the caller
        }
        double ln_fx = 0.0;
        if (bb.mode != EvalPlan::Beb::PARTIAL) {
            // ln f(X) = ln sum_g exp(lnfX_g) around the largest term (beb.py:609-611); posterior grid weights
            double s2 = -std::numeric_limits<double>::infinity(), sum = 0.0;
            for (int g = 0; g < bb.G; ++g) s2 = std::max(s2, h_lnfx[g]);
            for (int g = 0; g < bb.G; ++g) sum += std::exp(h_lnfx[g] - s2);
            ln_fx = std::log(sum) + s2;
            for (int g = 0; g < bb.G; ++g) h_w[g] = std::exp(h_lnfx[g] - ln_fx);
            HIP_CHECK(hipMemcpyAsync(e->d_beb_w.p, h_w, bb.G * sizeof(double), hipMemcpyHostToDevice, e->stream));
            launch_beb_posteriors(ba, e->stream);
            if (e->n_pat > 0)
                HIP_CHECK(hipMemcpyAsync(h_post, e->d_beb_post.p, (size_t)e->n_pat * bb.n_mix * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        }
        if (bb.lnfx && bb.mode != EvalPlan::Beb::GIVEN) std::copy(h_lnfx, h_lnfx + bb.G, bb.lnfx);
        if (bb.ln_fx) *bb.ln_fx = ln_fx;
    } else if (pl.by_class) {
        launch_by_class(e->d_f.p, e->d_e.p, pl.K, e->n_pat, e->n_pat_pad, pl.clamp ? 1 : 0, e->d_byclass.p, e->stream);
        HIP_CHECK(hipMemcpyAsync(e->h_out.p + G, e->d_byclass.p, (size_t)e->n_pat * pl.K * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    } else {
        MixArgs ma{};
        ma.f = e->d_f.p;
        ma.e = e->d_e.p;
        ma.weights = e->d_weights.p;
        ma.props = d_props;
        ma.K = pl.K / G;
        ma.n_groups = G;
        ma.n_pat = e->n_pat;
        ma.n_pat_pad = e->n_pat_pad;
        ma.mode = pl.mix_mode;
        ma.pat_lnl = e->d_patlnl.p;
        ma.partials = e->d_partials.p;
        // The G sums go straight into the pinned host buffer (the final-sum kernel's one 8-byte store per parameter vector
        // crosses PCIe itself): no device-to-host copy, which the runtime performs as a launch of its own (4.5 us of device
        // time plus its dispatch, per evaluation).  Visible to the host after the stream synchronisation below.
        static const bool no_zero_copy = std::getenv("CRAB_NO_ZERO_COPY") != nullptr;   // A/B and the bit-identity test
        const bool zero_copy = !no_zero_copy && e->h_out.dev != nullptr;
        ma.lnL = zero_copy ? e->h_out.dev : e->d_lnL.p;
        if (e->n_pat > 0) {
            launch_mix(ma, e->stream);
            if (!zero_copy) HIP_CHECK(hipMemcpyAsync(e->h_out.p, e->d_lnL.p, G * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            if (site_lnl)
                HIP_CHECK(hipMemcpyAsync(e->h_out.p + G, e->d_patlnl.p, e->n_pat * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        } else {
            for (int g = 0; g < G; ++g) e->h_out.p[g] = 0.0;
        }
    }
    HIP_CHECK(hipEventRecord(e->ev_end, e->stream));
    // host work that can hide behind the device: remember the decompositions made for this call
    for (size_t j = 0; j < todo.size(); ++j) {
        const int i = uniq_src[todo[j]];
        EigPool::Entry &en = pool.slots[new_slots[j]];
        en.used = true;
        en.by_params = by_params;
        en.h = qhash[i];
        en.h2 = qhash2[i];
        en.stamp = pool.clock;
        en.pi.assign(pi, pi + n);
        if (by_params) {
            en.kappa = pl.kw[i].first;
            en.omega = pl.kw[i].second;
        } else {
            if (EigPool::confirm_hits()) en.Q.assign(Q + (size_t)i * n * n, Q + (size_t)(i + 1) * n * n);
        }
        pool.index.emplace(en.h, new_slots[j]);
    }
    HIP_CHECK(hipStreamSynchronize(e->stream));
    HIP_CHECK(hipGetLastError());
    if (device_eigen && e->h_eig_status.p[0] != 0) {   // forget what kernel 0 left in the new slots
        for (int sl : new_slots) pool.drop(sl);
        fail(CRAB_ERR_EIGEN, "Eigendecomposition failed: the device solver did not converge in 24 sweeps");
    }
    if (merged && e->h_sync_flag.p && e->h_sync_flag.p[0] != 0) {
        // The merged clade launch left tables incomplete: a work item gave up waiting, or one of the device's XCDs ran no workgroup
        // of the launch (nothing promises how a device deals workgroups to its XCDs).  Nothing of this evaluation is used: the
        // engine goes back to one launch per nesting level, for good, and evaluates again (the decompositions are in the pool).
        if (e->merged_clade_launch_off) fail(CRAB_ERR_DEVICE, "clade tables incomplete");
        fprintf(stderr, "[crabml_amd] device %d: the one-launch schedule of the clade programs left tables incomplete; this engine "
                        "uses one launch per nesting level from here on (CRAB_CLADE_LEVEL_LAUNCHES=1 does so from the start)\n", e->device);
        e->merged_clade_launch_off = true;
        e->dev_ops.clear();   // plan and upload the programs again
        run_plan(e, pl, pi, lnL, site_lnl, by_class_out);
        return;
    }
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev_begin, e->ev_end));
    e->last_device_ms = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev_k0, e->ev_k1));
    e->last_prune_ms = ms;
    e->last_pidx = *pidx;
    e->last_K = pl.K;
    e->last_M = M;
    e->last_dense = e->dev_gemm_per_pat * e->n_pat + e->dev_gemm_extra;     // products and gathers actually executed
    e->last_gather = e->dev_leaf_per_pat * e->n_pat;
    e->last_n_path = n_parts > 0 ? 0 : n_second / std::max(1, Kg);
    e->last_n_parts = n_parts;
    e->last_n_clades = n_cl;
    e->last_n_launches = (use_clades ? (int)e->dev_clade_launches.size() : 0) + 1 + (two_launches ? 1 : 0);

    if (pl.post_weights) {
        const double *src = e->h_out.p + G;
        const int Kc = pl.K / G;
        for (int64_t s = 0; s < e->n_sites; ++s) {
            const int64_t p = e->site_to_pat[s];
            if (p < e->pat_begin || p >= e->pat_end) continue;
            std::memcpy(pl.post_out + (size_t)s * Kc, src + (size_t)(p - e->pat_begin) * Kc, Kc * sizeof(double));
        }
    } else if (pl.beb && pl.beb->mode == EvalPlan::Beb::PARTIAL) {
        // nothing per site: the caller sums the lnfX of all slices and comes back
    } else if (pl.beb) {
        const double *src = e->h_out.p + G + 2 * (size_t)pl.beb->G;
        const int nm = pl.beb->n_mix;
        for (int64_t s = 0; s < e->n_sites; ++s) {
            const int64_t p = e->site_to_pat[s];
            if (p < e->pat_begin || p >= e->pat_end) continue;
            std::memcpy(pl.beb->post + (size_t)s * nm, src + (size_t)(p - e->pat_begin) * nm, nm * sizeof(double));
        }
    } else if (pl.by_class) {
        const double *src = e->h_out.p + G;
        for (int64_t s = 0; s < e->n_sites; ++s) {
            const int64_t p = e->site_to_pat[s];
            if (p < e->pat_begin || p >= e->pat_end) continue;
            std::memcpy(by_class_out + (size_t)s * pl.K, src + (size_t)(p - e->pat_begin) * pl.K, pl.K * sizeof(double));
        }
    } else {
        if (lnL)
            for (int g = 0; g < G; ++g) lnL[g] = e->h_out.p[g];
        if (site_lnl) {
            const double *src = e->h_out.p + G;
            for (int64_t s = 0; s < e->n_sites; ++s) {
                const int64_t p = e->site_to_pat[s];
                if (p < e->pat_begin || p >= e->pat_end) continue;
                site_lnl[s] = src[p - e->pat_begin];
            }
        }
    }
}

void plan_site_classes(crab_engine *e, EvalPlan &pl, int K, const double *Q, const double *t)
{
    const int nn = e->n_nodes;
    pl.n_eig = K;
    pl.Q = Q;
    pl.K = K;
    pl.M = K * nn;
    pl.eig_id.resize(pl.M);
    pl.t_eff.resize(pl.M);
    pl.pidx.resize((size_t)K * nn);
    for (int k = 0; k < K; ++k)
        for (int v = 0; v < nn; ++v) {
            const int m = k * nn + v;
            pl.eig_id[m] = k;
            pl.t_eff[m] = v == e->root ? 0.0 : t[v];
            pl.pidx[m] = m;
        }
}


}  // namespace crab_host

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int crab_device_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n > 0 ? 1 : 0;
}

int crab_create(int device, crab_engine **out)
{
    if (!out) return CRAB_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<crab_engine> e(new (std::nothrow) crab_engine());
    if (!e) return CRAB_ERR_DEVICE;
    const int rc = guarded(nullptr, [&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            throw HipError{"no HIP device available: libcrabml_amd needs an AMD GPU (gfx950); there is no CPU fallback"};
        }
        if (device < 0 || device >= n) fail(CRAB_ERR_ARG, "device ordinal " + std::to_string(device) + " out of range");
        e->device = device;
        HIP_CHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            throw HipError{std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only"};
        HIP_CHECK((hipError_t)prune_configure_device());
        HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&e->ev_begin));
        HIP_CHECK(hipEventCreate(&e->ev_end));
        HIP_CHECK(hipEventCreate(&e->ev_k0));
        HIP_CHECK(hipEventCreate(&e->ev_k1));
        if (const char *sv = std::getenv("CRAB_DEVICE_EIGEN")) e->device_eigen = sv[0] && sv[0] != '0';
    });
    if (rc == CRAB_OK) *out = e.release();
    return rc;
}

int crab_create_multi(const int *devices, int n_devices, crab_engine **out)
{
    if (!out) return CRAB_ERR_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        g_last_error = "crab_create_multi: need 1 to 64 device ordinals";
        return CRAB_ERR_ARG;
    }
    std::unique_ptr<crab_engine> e(new (std::nothrow) crab_engine());
    if (!e) return CRAB_ERR_DEVICE;
    e->device = devices[0];
    for (int r = 0; r < n_devices; ++r) {
        crab_engine *sh = nullptr;
        const int rc = crab_create(devices[r], &sh);   // leaves its message in g_last_error
        if (rc != CRAB_OK) return rc;                  // e's destructor frees the shards made so far
        e->shards.push_back(sh);
    }
    e->shared_eig_owner.reset(new SharedEigCache());
    for (crab_engine *sh : e->shards) sh->shared_eig = e->shared_eig_owner.get();
    e->workers.reset(new ShardWorkers());
    e->workers->start(n_devices - 1);
    *out = e.release();
    return CRAB_OK;
}

int crab_set_eigensolver(crab_engine *e, int on_device)
{
    if (!e) return CRAB_ERR_ARG;
    e->device_eigen = on_device != 0;
    for (crab_engine *sh : e->shards) sh->device_eigen = on_device != 0;
    return CRAB_OK;
}

int crab_n_devices(const crab_engine *e) { return e ? (e->is_multi() ? (int)e->shards.size() : 1) : 0; }

int crab_set_clade_tables(crab_engine *e, int on)
{
    if (!e) return CRAB_ERR_ARG;
    e->clade_tables_on = on != 0;
    for (crab_engine *sh : e->shards) sh->clade_tables_on = on != 0;
    return CRAB_OK;
}

int crab_last_prune_launches(const crab_engine *e, int *n_launches)
{
    if (!e || !n_launches) return CRAB_ERR_ARG;
    *n_launches = e->is_multi() ? e->shards[0]->last_n_launches : e->last_n_launches;
    return CRAB_OK;
}

int crab_last_clade_tables(const crab_engine *e, int *n_clades, int64_t *sub_patterns)
{
    if (!e) return CRAB_ERR_ARG;
    int n = 0;
    int64_t u = 0;
    auto add = [&](const crab_engine *x) {
        n += x->last_n_clades;
        if (x->last_n_clades > 0)
            for (const auto &cl : x->clades.clades) u += cl.U;
    };
    if (e->is_multi())
        for (const crab_engine *sh : e->shards) add(sh);
    else
        add(e);
    if (n_clades) *n_clades = n;
    if (sub_patterns) *sub_patterns = u;
    return CRAB_OK;
}

void crab_destroy(crab_engine *e) { delete e; }

const char *crab_last_error(const crab_engine *e) { return e ? e->err.c_str() : g_last_error.c_str(); }

int crab_set_tree(crab_engine *e, int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids,
                  int n_leaves)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {
        const int rc = fan_out(e, [&](crab_engine *sh, int) { return crab_set_tree(sh, n_nodes, node_ids, parents, leaf_node_ids, n_leaves); });
        e->have_tree = rc == CRAB_OK;
        e->n_nodes = e->shards[0]->n_nodes;
        e->root = e->shards[0]->root;
        return rc;
    }
    return guarded(e, [&] {
        e->have_tree = false;
        e->dev_ops.clear();
        e->path_programs.clear();
        e->tree_parts = TreeParts{};
        e->clades = CladeTables{};
        build_tree(e, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        e->have_tree = true;
    });
}

int crab_set_alignment(crab_engine *e, const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress,
                       int64_t pattern_begin, int64_t pattern_end)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {
        // compress once, then every shard uploads its own tile-aligned slice of [pattern_begin, pattern_end)
        CompressedAlignment ca;
        e->have_aln = false;
        int rc = guarded(e, [&] { compress_alignment(codes, n_leaves, n_sites, n_states, compress, ca); });
        if (rc != CRAB_OK) return rc;
        const int64_t total = (int64_t)ca.first_site.size();
        int64_t lo = std::max<int64_t>(0, pattern_begin), hi = (pattern_end < 0 || pattern_end > total) ? total : pattern_end;
        if (lo > hi) lo = hi;
        const int N = (int)e->shards.size();
        const int64_t tile = prune_tile_width(), tiles = (hi - lo + tile - 1) / tile, per = (tiles + N - 1) / N;
        rc = fan_out(e, [&](crab_engine *sh, int r) {
            const int64_t b = std::min(hi, lo + (int64_t)r * per * tile), en = std::min(hi, lo + (int64_t)(r + 1) * per * tile);
            return guarded(sh, [&] { upload_alignment(sh, ca, b, en); });
        });
        if (rc != CRAB_OK) return rc;
        e->n_states = n_states;
        e->aln_rows = n_leaves;
        e->n_sites = n_sites;
        e->n_pat_total = total;
        e->pat_begin = lo;
        e->pat_end = hi;
        e->n_pat = hi - lo;
        e->have_aln = true;
        return CRAB_OK;
    }
    return guarded(e, [&] { load_alignment(e, codes, n_leaves, n_sites, n_states, compress, pattern_begin, pattern_end); });
}

int64_t crab_n_patterns(const crab_engine *e) { return e ? e->n_pat : 0; }
int64_t crab_n_patterns_total(const crab_engine *e) { return e ? e->n_pat_total : 0; }
int64_t crab_n_sites(const crab_engine *e) { return e ? e->n_sites : 0; }

int crab_eval_site_classes(crab_engine *e, int K, const double *Q, const double *proportions, const double *pi,
                           const double *branch_lengths, unsigned flags, double *lnL, double *site_lnl)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, 1, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_site_classes(sh, K, Q, proportions, pi, branch_lengths, flags, part, site_lnl);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (K < 1 || !Q || !proportions || !pi || !branch_lengths || !lnL) fail(CRAB_ERR_ARG, "crab_eval_site_classes: bad arguments");
        EvalPlan pl;
        plan_site_classes(e, pl, K, Q, branch_lengths);
        pl.props.assign(proportions, proportions + K);
        pl.mix_mode = (flags & CRAB_FLAG_NO_CLAMP) ? MIX_LOGSUMEXP_NOCLAMP : MIX_LOGSUMEXP_CLAMP;
        run_plan(e, pl, pi, lnL, site_lnl, nullptr);
    });
}

int crab_eval_site_classes_batch(crab_engine *e, int B, int K, const double *Q, const double *proportions, const double *pi,
                                 const double *branch_lengths, unsigned flags, double *lnL)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, B, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_site_classes_batch(sh, B, K, Q, proportions, pi, branch_lengths, flags, part);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (B < 1 || K < 1 || !Q || !proportions || !pi || !branch_lengths || !lnL)
            fail(CRAB_ERR_ARG, "crab_eval_site_classes_batch: bad arguments");
        const int nn = e->n_nodes;
        EvalPlan pl;
        pl.n_groups = B;
        pl.n_eig = B * K;
        pl.Q = Q;
        pl.K = B * K;
        pl.M = B * K * nn;
        pl.eig_id.resize(pl.M);
        pl.t_eff.resize(pl.M);
        pl.pidx.resize((size_t)pl.K * nn);
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                for (int v = 0; v < nn; ++v) {
                    const int m = (b * K + k) * nn + v;
                    pl.eig_id[m] = b * K + k;
                    pl.t_eff[m] = v == e->root ? 0.0 : branch_lengths[(size_t)b * nn + v];
                    pl.pidx[m] = m;
                }
        pl.props.assign(proportions, proportions + (size_t)B * K);
        pl.mix_mode = (flags & CRAB_FLAG_NO_CLAMP) ? MIX_LOGSUMEXP_NOCLAMP : MIX_LOGSUMEXP_CLAMP;
        run_plan(e, pl, pi, lnL, nullptr, nullptr);
    });
}

int crab_eval_site_lnl_by_class(crab_engine *e, int K, const double *Q, const double *pi, const double *branch_lengths,
                                unsigned flags, double *out)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())   // every shard writes the rows of its own sites
        return fan_out(e, [&](crab_engine *sh, int) { return crab_eval_site_lnl_by_class(sh, K, Q, pi, branch_lengths, flags, out); });
    return guarded(e, [&] {
        check_ready(e);
        if (K < 1 || !Q || !pi || !branch_lengths || !out) fail(CRAB_ERR_ARG, "crab_eval_site_lnl_by_class: bad arguments");
        EvalPlan pl;
        plan_site_classes(e, pl, K, Q, branch_lengths);
        pl.by_class = true;
        pl.clamp = !(flags & CRAB_FLAG_NO_CLAMP);
        run_plan(e, pl, pi, nullptr, nullptr, out);
    });
}

int crab_eval_beb_grid(crab_engine *e, int K, const double *Q, const double *pi, const double *branch_lengths, int G, int n_mix,
                       const int32_t *cols, const double *props, unsigned flags, double *posteriors, double *lnfx, double *ln_fx)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {
        // f(X | grid point) is a product over all sites: the shards' per-grid-point sums are added in shard order, then
        // every shard integrates the posteriors of its own sites (the protocol of CRAB_FLAG_BEB_PARTIAL / _GIVEN_LNFX)
        const bool partial = flags & CRAB_FLAG_BEB_PARTIAL, given = flags & CRAB_FLAG_BEB_GIVEN_LNFX;
        if (G < 1 || (partial && given) || ((partial || given) && !lnfx) || (!posteriors && !partial)) {
            e->err = "crab_eval_beb_grid: bad arguments";
            return CRAB_ERR_ARG;
        }
        if (!partial && !given && (e->pat_begin != 0 || e->pat_end != e->n_pat_total)) {
            e->err = "crab_eval_beb_grid needs the whole alignment on one handle, or the two-call protocol of "
                     "CRAB_FLAG_BEB_PARTIAL / CRAB_FLAG_BEB_GIVEN_LNFX";
            return CRAB_ERR_STATE;
        }
        const int N = (int)e->shards.size();
        std::vector<double> total(G, 0.0);
        if (!given) {
            std::vector<double> part((size_t)N * G, 0.0);
            const int rc = fan_out(e, [&](crab_engine *sh, int r) {
                return crab_eval_beb_grid(sh, K, Q, pi, branch_lengths, G, n_mix, cols, props, CRAB_FLAG_BEB_PARTIAL, nullptr,
                                          part.data() + (size_t)r * G, nullptr);
            });
            if (rc != CRAB_OK) return rc;
            for (int g = 0; g < G; ++g)
                for (int r = 0; r < N; ++r) total[g] += part[(size_t)r * G + g];
            if (lnfx) std::copy(total.begin(), total.end(), lnfx);
            if (partial) {
                if (ln_fx) *ln_fx = 0.0;
                return CRAB_OK;
            }
        } else {
            std::copy(lnfx, lnfx + G, total.begin());
        }
        std::vector<double> lf(N, 0.0);
        const int rc = fan_out(e, [&](crab_engine *sh, int r) {
            return crab_eval_beb_grid(sh, K, Q, pi, branch_lengths, G, n_mix, cols, props, CRAB_FLAG_BEB_GIVEN_LNFX, posteriors,
                                      total.data(), &lf[r]);
        });
        if (rc == CRAB_OK && ln_fx) *ln_fx = lf[0];
        return rc;
    }
    return guarded(e, [&] {
        check_ready(e);
        const bool partial = flags & CRAB_FLAG_BEB_PARTIAL, given = flags & CRAB_FLAG_BEB_GIVEN_LNFX;
        if (K < 1 || !Q || !pi || !branch_lengths || G < 1 || !cols || !props || (!posteriors && !partial) || (partial && given) ||
            ((partial || given) && !lnfx))
            fail(CRAB_ERR_ARG, "crab_eval_beb_grid: bad arguments");
        if (K > NS - 1) fail(CRAB_ERR_ARG, "crab_eval_beb_grid: at most 63 classes");
        if (n_mix < 1 || n_mix > BEB_MAX_MIX) fail(CRAB_ERR_ARG, "crab_eval_beb_grid: n_mix must be in [1, 16]");
        for (size_t i = 0; i < (size_t)G * n_mix; ++i)
            if (cols[i] < 0 || cols[i] >= K) fail(CRAB_ERR_ARG, "crab_eval_beb_grid: class index out of range in the grid table");
        if (!partial && !given && (e->pat_begin != 0 || e->pat_end != e->n_pat_total))
            fail(CRAB_ERR_STATE, "crab_eval_beb_grid needs the whole alignment on one engine (f(X | grid point) is a product over all "
                                 "sites), or the two-call protocol of CRAB_FLAG_BEB_PARTIAL / CRAB_FLAG_BEB_GIVEN_LNFX");
        EvalPlan pl;
        plan_site_classes(e, pl, K, Q, branch_lengths);
        EvalPlan::Beb bb;
        bb.G = G;
        bb.n_mix = n_mix;
        bb.cols = cols;
        bb.props = props;
        bb.post = posteriors;
        bb.lnfx = lnfx;
        bb.ln_fx = ln_fx;
        bb.mode = partial ? EvalPlan::Beb::PARTIAL : (given ? EvalPlan::Beb::GIVEN : EvalPlan::Beb::WHOLE);
        pl.beb = &bb;
        run_plan(e, pl, pi, nullptr, nullptr, nullptr);
    });
}

int crab_eval_branch_site(crab_engine *e, const double *Q3, double qfactor_back, double qfactor_fore, double omega0,
                          double omega2, const double *site_class_freqs, const double *pi, const double *branch_lengths,
                          const uint8_t *labels, int n_labels, unsigned flags, double *lnL, double *site_lnl)
{
    (void)flags;
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, 1, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_branch_site(sh, Q3, qfactor_back, qfactor_fore, omega0, omega2, site_class_freqs, pi, branch_lengths, labels,
                                         n_labels, flags, part, site_lnl);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (!Q3 || !site_class_freqs || !pi || !branch_lengths || !labels || !lnL) fail(CRAB_ERR_ARG, "crab_eval_branch_site: bad arguments");
        const int nn = e->n_nodes;
        if (n_labels != nn - 1)  // branch_site_likelihood.rs:41-47
            fail(CRAB_ERR_ARG, "Branch labels length " + std::to_string(n_labels) + " must equal n_nodes - 1 = " + std::to_string(nn - 1));
        for (int v = 0; v < n_labels; ++v)
            if (labels[v] > 1) fail(CRAB_ERR_ARG, "Invalid branch_label or site_class");  // the reference panics (:100)
        // eigensystem per (label, class): select_omega (:85-102) followed by the float comparisons of :195-201
        auto pick = [&](int lab, int k) {
            const double w = k == 0 ? omega0 : k == 1 ? 1.0 : (lab == 0 ? (k == 2 ? omega0 : 1.0) : omega2);
            return w == omega0 ? 0 : (w == 1.0 ? 1 : 2);
        };
        EvalPlan pl;
        pl.n_eig = 3;
        pl.Q = Q3;
        pl.K = 4;
        pl.pidx.assign((size_t)4 * nn, 0);
        pl.props.assign(site_class_freqs, site_class_freqs + 4);
        pl.mix_mode = MIX_LINEAR;
        for (int v = 0; v < nn; ++v) {
            const int lab = v < n_labels ? labels[v] : 0;  // root and beyond: background (:61-70)
            const double t = (v == e->root ? 0.0 : branch_lengths[v]) * (lab == 0 ? qfactor_back : qfactor_fore);
            int slot_of_eig[3] = {-1, -1, -1};
            for (int k = 0; k < 4; ++k) {
                const int q = pick(lab, k);
                if (slot_of_eig[q] < 0) {
                    slot_of_eig[q] = pl.M++;
                    pl.eig_id.push_back(q);
                    pl.t_eff.push_back(t);
                }
                pl.pidx[(size_t)k * nn + v] = slot_of_eig[q];
            }
        }
        run_plan(e, pl, pi, lnL, site_lnl, nullptr);
    });
}

int crab_eval_branch(crab_engine *e, int n_q, const double *Q, const double *pi, const double *branch_lengths,
                     int n_branch_lengths, unsigned flags, double *lnL, double *site_lnl)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi())
        return multi_sum(e, 1, lnL, [&](crab_engine *sh, double *part) {
            return crab_eval_branch(sh, n_q, Q, pi, branch_lengths, n_branch_lengths, flags, part, site_lnl);
        });
    return guarded(e, [&] {
        check_ready(e);
        if (!Q || !pi || !branch_lengths || !lnL) fail(CRAB_ERR_ARG, "crab_eval_branch: bad arguments");
        if (n_q != n_branch_lengths)  // lib.rs:377-384
            fail(CRAB_ERR_ARG, "Q_matrices length (" + std::to_string(n_q) + ") must match branch_lengths length (" +
                                   std::to_string(n_branch_lengths) + ")");
        if (n_q != e->n_nodes)  // the reference panics (likelihood.rs:392-398)
            fail(CRAB_ERR_ARG, "Number of Q matrices (" + std::to_string(n_q) + ") must equal number of nodes (" +
                                   std::to_string(e->n_nodes) + ")");
        const int nn = e->n_nodes;
        EvalPlan pl;
        pl.n_eig = nn;
        pl.Q = Q;
        pl.K = 1;
        pl.M = nn;
        pl.eig_id.resize(nn);
        pl.t_eff.resize(nn);
        pl.pidx.resize(nn);
        for (int v = 0; v < nn; ++v) {
            pl.eig_id[v] = v;
            pl.t_eff[v] = v == e->root ? 0.0 : branch_lengths[v];
            pl.pidx[v] = v;
        }
        pl.props.assign(1, 1.0);
        pl.mix_mode = (flags & CRAB_FLAG_NO_CLAMP) ? MIX_LOGSUMEXP_NOCLAMP : MIX_LOGSUMEXP_CLAMP;
        run_plan(e, pl, pi, lnL, site_lnl, nullptr);
    });
}

int crab_last_timing(const crab_engine *e, double *device_ms, double *prune_kernel_ms, double *host_eigen_ms)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {   // the shards run side by side: the slowest one is the call
        double d = 0, p = 0, h = 0;
        for (const crab_engine *sh : e->shards) {
            d = std::max(d, sh->last_device_ms);
            p = std::max(p, sh->last_prune_ms);
            h = std::max(h, sh->last_eigen_ms);
        }
        if (device_ms) *device_ms = d;
        if (prune_kernel_ms) *prune_kernel_ms = p;
        if (host_eigen_ms) *host_eigen_ms = h;
        return CRAB_OK;
    }
    if (device_ms) *device_ms = e->last_device_ms;
    if (prune_kernel_ms) *prune_kernel_ms = e->last_prune_ms;
    if (host_eigen_ms) *host_eigen_ms = e->last_eigen_ms;
    return CRAB_OK;
}

int crab_export_transition_matrices(crab_engine *e, int k, double *P_out)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {   // every shard built the same matrices
        const int rc = crab_export_transition_matrices(e->shards[0], k, P_out);
        if (rc != CRAB_OK) e->err = e->shards[0]->err;
        return rc;
    }
    return guarded(e, [&] {
        if (!P_out || k < 0 || k >= e->last_K || e->last_M <= 0) fail(CRAB_ERR_STATE, "no evaluation to export from, or class out of range");
        const int n = e->n_states, nn = e->n_nodes;
        HIP_CHECK(hipSetDevice(e->device));
        std::vector<double> img((size_t)e->last_M * P_ELEMS);
        HIP_CHECK(hipMemcpy(img.data(), e->d_P.p, img.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int v = 0; v < nn; ++v) {
            const double *Pm = img.data() + (size_t)e->last_pidx[(size_t)k * nn + v] * P_ELEMS;
            double *out = P_out + (size_t)v * n * n;
            const bool gathered = v != e->root && e->reachable[v] && e->in_leaf_set[v];   // a leaf edge has its gather image only
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = Pm[gathered ? p_gather_index(i, j) : p_operand_index(i, j)];
        }
    });
}

int crab_eig_pool_stats(const crab_engine *e, int64_t *hits, int64_t *misses, int64_t *resident)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {   // sums over the shards' device-resident pools
        int64_t h = 0, m = 0, r = 0;
        for (const crab_engine *sh : e->shards) {
            int64_t a = 0, b = 0, c = 0;
            crab_eig_pool_stats(sh, &a, &b, &c);
            h += a, m += b, r += c;
        }
        if (hits) *hits = h;
        if (misses) *misses = m;
        if (resident) *resident = r;
        return CRAB_OK;
    }
    if (hits) *hits = e->eig_hits;
    if (misses) *misses = e->eig_misses;
    if (resident) {
        int64_t n = 0;
        for (const auto &en : e->eig_pool.slots) n += en.used ? 1 : 0;
        *resident = n;
    }
    return CRAB_OK;
}

int crab_last_partial_updates(const crab_engine *e, int *n_vectors)
{
    if (!e || !n_vectors) return CRAB_ERR_ARG;
    *n_vectors = e->is_multi() ? e->shards[0]->last_n_path : e->last_n_path;
    return CRAB_OK;
}

int crab_last_tree_parts(const crab_engine *e, int *n_parts)
{
    if (!e || !n_parts) return CRAB_ERR_ARG;
    *n_parts = e->is_multi() ? e->shards[0]->last_n_parts : e->last_n_parts;
    return CRAB_OK;
}

int crab_last_work(const crab_engine *e, int64_t *dense_updates, int64_t *gather_updates)
{
    if (!e) return CRAB_ERR_ARG;
    if (e->is_multi()) {
        int64_t d = 0, g = 0;
        for (const crab_engine *sh : e->shards) d += sh->last_dense, g += sh->last_gather;
        if (dense_updates) *dense_updates = d;
        if (gather_updates) *gather_updates = g;
        return CRAB_OK;
    }
    if (dense_updates) *dense_updates = e->last_dense;
    if (gather_updates) *gather_updates = e->last_gather;
    return CRAB_OK;
}

// ---- host-only introspection (no device needed) ------------------------------------------------
int crab_compile_tree(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                      int32_t *ops_out, int ops_capacity, int *n_ops, int *max_depth)
{
    return guarded(nullptr, [&] {
        TreeProgram tp;
        build_tree(&tp, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        if (n_ops) *n_ops = (int)tp.ops.size();
        if (max_depth) *max_depth = tp.max_depth;
        if (ops_out) {
            if ((int)tp.ops.size() > ops_capacity) fail(CRAB_ERR_ARG, "ops_capacity too small");
            for (size_t i = 0; i < tp.ops.size(); ++i) {
                const TreeOp &o = tp.ops[i];
                int32_t *r = ops_out + 5 * i;
                r[0] = o.kind; r[1] = o.node; r[2] = o.flags; r[3] = o.arg; r[4] = o.slot;
            }
        }
    });
}

int crab_compile_path_program(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids,
                              int n_leaves, int node, int32_t *ops_out, int ops_capacity, int *n_ops, int *n_save_slots)
{
    return guarded(nullptr, [&] {
        TreeProgram tp;
        build_tree(&tp, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        const std::vector<TreeOp> *ops = &tp.ops;
        PathProgram pp;
        if (node >= 0) {
            if (node >= n_nodes || node == tp.root || !tp.reachable[node]) fail(CRAB_ERR_ARG, "crab_compile_path_program: no branch above this node");
            build_path_program(&tp, node, pp);
            ops = &pp.ops;
        }
        if (n_ops) *n_ops = (int)ops->size();
        if (n_save_slots) *n_save_slots = tp.n_save_slots;
        if (ops_out) {
            if ((int)ops->size() > ops_capacity) fail(CRAB_ERR_ARG, "ops_capacity too small");
            for (size_t i = 0; i < ops->size(); ++i) {
                const TreeOp &o = (*ops)[i];
                int32_t *r = ops_out + 6 * i;
                r[0] = o.kind; r[1] = o.node; r[2] = o.flags; r[3] = o.arg; r[4] = o.slot; r[5] = o.sv;
            }
        }
    });
}

int crab_compile_tree_part(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                           int target, int part, int32_t *ops_out, int ops_capacity, int *n_ops, int *n_parts)
{
    return guarded(nullptr, [&] {
        if (target < 1) fail(CRAB_ERR_ARG, "crab_compile_tree_part: target must be >= 1");
        TreeProgram tp;
        build_tree(&tp, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        TreeParts parts;
        build_tree_parts(&tp, target, parts);
        if (n_parts) *n_parts = (int)parts.cut.size();
        if (part >= (int)parts.cut.size()) fail(CRAB_ERR_ARG, "crab_compile_tree_part: no such part");
        const std::vector<TreeOp> &ops = part < 0 ? parts.root.ops : parts.parts[part].ops;
        if (n_ops) *n_ops = (int)ops.size();
        if (ops_out) {
            if ((int)ops.size() > ops_capacity) fail(CRAB_ERR_ARG, "ops_capacity too small");
            for (size_t i = 0; i < ops.size(); ++i) {
                const TreeOp &o = ops[i];
                int32_t *r = ops_out + 6 * i;
                r[0] = o.kind; r[1] = o.node; r[2] = o.flags; r[3] = o.arg; r[4] = o.slot; r[5] = o.sv;
            }
        }
    });
}

int crab_compile_clade_tables(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                              const int32_t *codes, int64_t n_patterns, int n_states, int part, int32_t *ops_out, int ops_capacity,
                              int *n_ops, int *n_clades, int *clade_root, int *parent_clade, int *wide_row, int64_t *n_sub_patterns,
                              uint32_t *index_out, int32_t *sub_codes_out)
{
    return guarded(nullptr, [&] {
        if (!codes || n_patterns < 1 || n_states < 1 || n_states > NS - 1) fail(CRAB_ERR_ARG, "crab_compile_clade_tables: bad arguments");
        TreeProgram tp;
        build_tree(&tp, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        const int tile = prune_tile_width();
        const int64_t npp = (n_patterns + tile - 1) / tile * tile;
        std::vector<uint8_t> img((size_t)n_leaves * npp, (uint8_t)n_states);
        for (int r = 0; r < n_leaves; ++r)
            for (int64_t p = 0; p < n_patterns; ++p) {
                const int32_t c = codes[(size_t)r * n_patterns + p];
                img[(size_t)r * npp + p] = (c >= 0 && c < n_states) ? (uint8_t)c : (uint8_t)n_states;
            }
        CladeTables ct;
        std::vector<uint32_t> idx;
        std::vector<uint8_t> alt;
        plan_clade_tables(&tp, img.data(), n_leaves, n_patterns, npp, n_states, ct, idx, alt);
        const int n_cl = (int)ct.clades.size();
        if (n_clades) *n_clades = n_cl;
        if (part >= n_cl) fail(CRAB_ERR_ARG, "crab_compile_clade_tables: no such clade");
        const std::vector<TreeOp> &ops = part < 0 ? ct.main.ops : ct.clades[part].prog.ops;
        if (n_ops) *n_ops = (int)ops.size();
        if (ops_out) {
            if ((int)ops.size() > ops_capacity) fail(CRAB_ERR_ARG, "ops_capacity too small");
            for (size_t i = 0; i < ops.size(); ++i) {
                const TreeOp &o = ops[i];
                int32_t *r = ops_out + 6 * i;
                r[0] = o.kind; r[1] = o.node; r[2] = o.flags; r[3] = o.arg; r[4] = o.slot; r[5] = o.sv;
            }
        }
        if (part >= 0) {
            const CladeTables::Clade &cl = ct.clades[part];
            if (clade_root) *clade_root = cl.root;
            if (parent_clade) *parent_clade = cl.parent;
            if (wide_row) *wide_row = cl.row;
            if (n_sub_patterns) *n_sub_patterns = cl.U;
            if (index_out) {   // per pattern (consumer: the main program) or per distinct sub-pattern of the consuming clade
                const int64_t n = cl.parent < 0 ? n_patterns : ct.clades[cl.parent].U;
                const uint32_t *row = idx.data() + (size_t)cl.row * npp;
                for (int64_t p = 0; p < n; ++p) index_out[p] = row[p];
            }
            if (sub_codes_out) {   // [n_leaves][U]: the codes of the clade's distinct sub-patterns on its own rows, n_states elsewhere
                const uint8_t *lvl = alt.data() + (size_t)(cl.level - 1) * n_leaves * npp;
                for (int r = 0; r < n_leaves; ++r)
                    for (int64_t u = 0; u < cl.U; ++u) sub_codes_out[(size_t)r * cl.U + u] = lvl[(size_t)r * npp + u];
            }
        }
    });
}

int crab_plan_clade_work(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                         const int32_t *codes, int64_t n_patterns, int n_states, int *n_clades, int64_t *sub_patterns,
                         int64_t *products, int64_t *whole_walk_products)
{
    return guarded(nullptr, [&] {
        if (!codes || n_patterns < 1 || n_states < 1 || n_states > NS - 1) fail(CRAB_ERR_ARG, "crab_plan_clade_work: bad arguments");
        TreeProgram tp;
        build_tree(&tp, n_nodes, node_ids, parents, leaf_node_ids, n_leaves);
        const int tile = prune_tile_width();
        const int64_t npp = (n_patterns + tile - 1) / tile * tile;
        std::vector<uint8_t> img((size_t)n_leaves * npp, (uint8_t)n_states);
        for (int r = 0; r < n_leaves; ++r)
            for (int64_t p = 0; p < n_patterns; ++p) {
                const int32_t c = codes[(size_t)r * n_patterns + p];
                img[(size_t)r * npp + p] = (c >= 0 && c < n_states) ? (uint8_t)c : (uint8_t)n_states;
            }
        CladeTables ct;
        std::vector<uint32_t> idx;
        std::vector<uint8_t> alt;
        plan_clade_tables(&tp, img.data(), n_leaves, n_patterns, npp, n_states, ct, idx, alt);
        int64_t sub = 0, prod = 0;
        for (const auto &cl : ct.clades) {
            sub += cl.U;
            prod += (int64_t)cl.prog.info.n_gemm * cl.tiles * tile;
        }
        prod += (ct.clades.empty() ? tp.n_gemm_edges : (int64_t)ct.main.info.n_gemm) * n_patterns;
        if (n_clades) *n_clades = (int)ct.clades.size();
        if (sub_patterns) *sub_patterns = sub;
        if (products) *products = prod;
        if (whole_walk_products) *whole_walk_products = tp.n_gemm_edges * n_patterns;
    });
}

int crab_compress_patterns(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int64_t *site_to_pattern,
                           double *pattern_weight, int64_t *first_site, int64_t *n_patterns)
{
    return guarded(nullptr, [&] {
        if (!codes || n_leaves < 1 || n_sites < 0 || n_states < 1 || n_states > NS - 1 || !n_patterns)
            fail(CRAB_ERR_ARG, "crab_compress_patterns: bad arguments");
        std::vector<uint8_t> cols;
        std::vector<int64_t> s2p, first;
        std::vector<double> cnt;
        compress_patterns(codes, n_leaves, n_sites, n_states, 1, cols, s2p, first, cnt);
        *n_patterns = (int64_t)first.size();
        if (site_to_pattern) std::copy(s2p.begin(), s2p.end(), site_to_pattern);
        if (pattern_weight) std::copy(cnt.begin(), cnt.end(), pattern_weight);
        if (first_site) std::copy(first.begin(), first.end(), first_site);
    });
}

// Exercises the host worker pool the way concurrent engines do: `callers` threads issue `rounds` parallel loops of `n`
// indices each at the same time; every index of every loop must run exactly once.  Returns CRAB_OK or CRAB_ERR_STATE.
int crab_host_pool_selftest(int callers, int n, int rounds)
{
    return guarded(nullptr, [&] {
        if (callers < 1 || n < 1 || rounds < 1) fail(CRAB_ERR_ARG, "crab_host_pool_selftest: bad arguments");
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        for (int c = 0; c < callers; ++c)
            th.emplace_back([&, c] {
                std::vector<int> hits(n);
                for (int r = 0; r < rounds; ++r) {
                    std::fill(hits.begin(), hits.end(), 0);
                    volatile double sink = 0.0;
                    parallel_for(n, [&](int i) {
                        for (int k = 0; k < 200 * ((i + c + r) % 7); ++k) sink = sink + k;   // uneven work
                        ++hits[i];   // each index belongs to exactly one thread
                    });
                    for (int i = 0; i < n; ++i)
                        if (hits[i] != 1) ++bad;
                }
            });
        for (auto &t : th) t.join();
        if (bad.load() != 0) fail(CRAB_ERR_STATE, "host pool ran an index " + std::to_string(bad.load()) + " times too many or too few");
    });
}

int crab_spectral_decompose(int n, const double *Q, const double *pi, double *U, double *lam, double *Uinv)
{
    return guarded(nullptr, [&] {
        if (!Q || !pi || !U || !lam || !Uinv || n < 1) fail(CRAB_ERR_ARG, "crab_spectral_decompose: bad arguments");
        if (!spectral_decompose_plain(n, Q, pi, U, lam, Uinv)) fail(CRAB_ERR_EIGEN, "Eigendecomposition failed");
    });
}

// ---- stateless utilities ---------------------------------------------------------------------
int crab_matrix_exponential(int device, int n, const double *Q, const double *pi, double t, double *P_out)
{
    return guarded(nullptr, [&] {
        if (!Q || !pi || !P_out) fail(CRAB_ERR_ARG, "crab_matrix_exponential: bad arguments");
        if (n < 1 || n > NS - 1) fail(CRAB_ERR_ARG, "n must be in [1, 63], got " + std::to_string(n));
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) {
            (void)hipGetLastError();
            throw HipError{"no HIP device available: libcrabml_amd needs an AMD GPU (gfx950); there is no CPU fallback"};
        }
        HIP_CHECK(hipSetDevice(device));
        std::vector<double> eig(EIG_STRIDE), Pf(P_ELEMS);
        if (!spectral_decompose(n, Q, pi, eig.data())) fail(CRAB_ERR_EIGEN, "Eigendecomposition failed");
        DevBuf<double> d_eig, d_t, d_P;
        DevBuf<int32_t> d_id;
        struct Cleanup {
            DevBuf<double> &a, &b, &c;
            DevBuf<int32_t> &d;
            ~Cleanup() { a.release(); b.release(); c.release(); d.release(); }
        } cleanup{d_eig, d_t, d_P, d_id};
        d_eig.ensure(EIG_STRIDE);
        d_t.ensure(1);
        d_P.ensure(P_ELEMS);
        d_id.ensure(1);
        const int32_t zero = 0;
        HIP_CHECK(hipMemcpy(d_eig.p, eig.data(), EIG_STRIDE * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_t.p, &t, sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_id.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice));
        launch_build_p(d_eig.p, d_id.p, d_t.p, d_P.p, 1, n, false, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(Pf.data(), d_P.p, P_ELEMS * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) P_out[(size_t)i * n + j] = Pf[p_gather_index(i, j)];
    });
}


}  // extern "C"
