// Internal to the host side of libcrabml_amd.so (not installed): the engine's state and the functions its source files share.
//   crab_engine.cpp     one evaluation (run_plan) and the handle-based C ABI
//   crab_programs.cpp   tree -> device programs: whole walk, path programs, tree parts, clade tables
//   crab_workplan.cpp   how one evaluation's work is cut into launches and work items; uploads the programs
//   crab_alignment.cpp  site patterns: compression, shard order, upload; content digests
//   crab_models.cpp     rate matrices and model-level entry points (parameters in, likelihoods out)
//   crab_oneshot.cpp    the reference's stateless signatures over a cache of handles
#pragma once
#include "../../include/crabml_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <thread>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "crab_eigen.h"
#include "crab_quantiles.h"
#include "crab_kernels.h"



using namespace crab;

namespace crab_host {


extern thread_local std::string g_last_error;   // crab_last_error(NULL): errors of calls without a handle

struct HipError {
    std::string msg;
};

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _err = (expr);                                                                         \
        if (_err != hipSuccess) {                                                                         \
            throw HipError{std::string(#expr) + ": " + hipGetErrorString(_err)};                          \
        }                                                                                                 \
    } while (0)

struct ArgError {
    int code;
    std::string msg;
};

// growable device allocation; contents are not preserved across growth
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 8;
        HIP_CHECK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// growable pinned host staging buffer
template <typename T>
struct PinBuf {
    T *p = nullptr;
    T *dev = nullptr;   // the same memory as kernels address it (pinned host memory is mapped into the device's address space), or null
    size_t cap = 0;
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr;
        dev = nullptr;
        cap = 0;
        const size_t want = n + n / 8;
        HIP_CHECK(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) dev = static_cast<T *>(d);
        else (void)hipGetLastError();
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
    }
};

inline double now_ms()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}


}  // namespace crab_host
using namespace crab_host;

uint64_t hash_words(const void *p, size_t bytes, uint64_t seed);
void hash_words2(const void *p, size_t bytes, uint64_t seed, uint64_t out[2]);

// Device-resident pool of spectral decompositions.  Finite-difference gradients perturb one parameter at a time, and a
// branch-length perturbation changes no rate matrix at all, so most of the K (site classes) or n_nodes (free-ratio)
// decompositions of a call repeat those of the call before (lib.rs:163-169,387-393 redo all of them).  An entry is
// keyed either by the contents of (Q, pi) or -- for the entry points that take model parameters -- by (kappa, omega, pi)
// and then holds the decomposition of the *unnormalised* matrix: every normalisation the models apply is a scalar that
// folds into the branch length (exp((Q / mu) t) = exp(Q (t / mu))), so a change of class proportions touches no
// eigensystem.  Hits cost neither a host copy nor an upload: kernel 1 reads the pool through slot indices.
struct EigPool {
    struct Entry {
        bool used = false, by_params = false;
        uint64_t h = 0, h2 = 0, stamp = 0;   // (h, h2): two 64-bit digests of one pass over Q (content keys)
        double kappa = 0, omega = 0;
        std::vector<double> Q, pi;   // Q only for content keys, and only kept when hits are confirmed byte for byte
    };
    static constexpr size_t SOFT_MAX = 4096;  // x 66 KB = 270 MB of HBM; least recently used entries make room beyond it
    std::vector<Entry> slots;
    std::unordered_multimap<uint64_t, int> index;
    std::vector<int> free_list;
    double *d = nullptr;
    size_t cap = 0;
    uint64_t clock = 0;

    // A matrix passed by value is recognised by the 128 bits of (h, h2) and by pi -- as the one-shot functions recognise an
    // alignment.  Round 2 confirmed every hit byte for byte against a host copy of the matrix: for the 1 998 matrices of a
    // free-ratio call that meant reading 59 MB twice (hash, then compare) per evaluation, 1.1 ms of a 1.8 ms call whose
    // device part takes 0.6.  CRAB_EIG_CONFIRM=1 brings the comparison back.
    static bool confirm_hits()
    {
        static const bool on = [] {
            const char *sv = std::getenv("CRAB_EIG_CONFIRM");
            return sv && sv[0] && sv[0] != '0';
        }();
        return on;
    }
    int find_content(uint64_t h, uint64_t h2, const double *Q, const double *pi, int n)
    {
        auto r = index.equal_range(h);
        for (auto it = r.first; it != r.second; ++it) {
            Entry &en = slots[it->second];
            if (!en.by_params && en.h2 == h2 && (!confirm_hits() || std::memcmp(en.Q.data(), Q, sizeof(double) * n * n) == 0) &&
                std::memcmp(en.pi.data(), pi, sizeof(double) * n) == 0) {
                en.stamp = clock;
                return it->second;
            }
        }
        return -1;
    }
    int find_params(uint64_t h, double kappa, double omega, const double *pi, int n)
    {
        auto r = index.equal_range(h);
        for (auto it = r.first; it != r.second; ++it) {
            Entry &en = slots[it->second];
            if (en.by_params && std::memcmp(&en.kappa, &kappa, 8) == 0 && std::memcmp(&en.omega, &omega, 8) == 0 &&
                std::memcmp(en.pi.data(), pi, sizeof(double) * n) == 0) {
                en.stamp = clock;
                return it->second;
            }
        }
        return -1;
    }
    void drop(int slot)
    {
        Entry &en = slots[slot];
        auto r = index.equal_range(en.h);
        for (auto it = r.first; it != r.second; ++it)
            if (it->second == slot) {
                index.erase(it);
                break;
            }
        en = Entry{};
    }
    // n slots for this call's new decompositions: free ones first, then growth up to SOFT_MAX, then the least recently
    // used entries that this call does not use itself, then growth again
    void allocate(int n, std::vector<int> &out, hipStream_t stream)
    {
        out.clear();
        while ((int)out.size() < n && !free_list.empty()) {
            out.push_back(free_list.back());
            free_list.pop_back();
        }
        int need = n - (int)out.size();
        if (need > 0 && cap >= SOFT_MAX) {
            std::vector<std::pair<uint64_t, int>> old;
            for (size_t i = 0; i < cap; ++i)
                if (slots[i].used && slots[i].stamp != clock) old.emplace_back(slots[i].stamp, (int)i);
            std::sort(old.begin(), old.end());
            for (size_t i = 0; i < old.size() && need > 0; ++i, --need) {
                drop(old[i].second);
                out.push_back(old[i].second);
            }
        }
        if (need > 0) {
            const size_t want = std::max<size_t>({cap + need, std::min<size_t>(2 * cap, SOFT_MAX), 32});
            double *nd = nullptr;
            HIP_CHECK(hipMalloc((void **)&nd, want * crab::EIG_STRIDE * sizeof(double)));
            if (d) {
                hipError_t err = hipMemcpyAsync(nd, d, cap * crab::EIG_STRIDE * sizeof(double), hipMemcpyDeviceToDevice, stream);
                if (err == hipSuccess) err = hipStreamSynchronize(stream);
                if (err != hipSuccess) {
                    (void)hipFree(nd);
                    throw HipError{std::string("growing the eigensystem pool: ") + hipGetErrorString(err)};
                }
                (void)hipFree(d);
            }
            d = nd;
            slots.resize(want);
            for (size_t i = want; i-- > cap;) free_list.push_back((int)i);   // lowest index handed out first
            cap = want;
            while ((int)out.size() < n) {
                out.push_back(free_list.back());
                free_list.pop_back();
            }
        }
    }
    void give_back(const std::vector<int> &got)
    {
        for (int sl : got)
            if (!slots[sl].used) free_list.push_back(sl);
    }
    void release()
    {
        if (d) (void)hipFree(d);
        d = nullptr;
        cap = 0;
        slots.clear();
        index.clear();
        free_list.clear();
    }
};

// Host-side cache of packed decompositions shared by the engines of one multi-device handle (crab_create_multi): every
// device needs every eigensystem, so the first shard to ask for one computes it and the others copy it (an 8-GPU handle
// would otherwise decompose each rate matrix eight times on the same host cores).  Keyed like EigPool.
struct SharedEigCache {
    struct Item {
        bool by_params = false, ready = false, failed = false;
        double kappa = 0, omega = 0;
        std::vector<double> Q, pi, packed;
        uint64_t stamp = 0;
    };
    static constexpr size_t MAX_ITEMS = 4096;   // x 66 KB = 270 MB of host memory
    std::mutex m;
    std::condition_variable cv;
    std::unordered_multimap<uint64_t, std::shared_ptr<Item>> map;
    uint64_t clock = 0;

    // Copies the decomposition for this key into out (EIG_STRIDE doubles), running compute(out) if nobody has yet.
    bool get(uint64_t h, bool by_params, double kappa, double omega, const double *Q, const double *pi, int n, double *out,
             const std::function<bool(double *)> &compute)
    {
        std::shared_ptr<Item> it;
        bool mine = false;
        {
            std::unique_lock<std::mutex> lk(m);
            auto r = map.equal_range(h);
            for (auto i = r.first; i != r.second; ++i) {
                Item &c = *i->second;
                if (c.by_params != by_params || std::memcmp(c.pi.data(), pi, sizeof(double) * n) != 0) continue;
                if (by_params ? (std::memcmp(&c.kappa, &kappa, 8) == 0 && std::memcmp(&c.omega, &omega, 8) == 0)
                              : std::memcmp(c.Q.data(), Q, sizeof(double) * n * n) == 0) {
                    it = i->second;
                    break;
                }
            }
            if (!it) {
                it = std::make_shared<Item>();
                it->by_params = by_params;
                it->kappa = kappa;
                it->omega = omega;
                it->pi.assign(pi, pi + n);
                if (!by_params) it->Q.assign(Q, Q + (size_t)n * n);
                map.emplace(h, it);
                mine = true;
                if (map.size() > MAX_ITEMS) evict_locked();
            }
            it->stamp = ++clock;
            if (!mine) {
                cv.wait(lk, [&] { return it->ready || it->failed; });
                if (it->failed) return false;
                std::memcpy(out, it->packed.data(), sizeof(double) * crab::EIG_STRIDE);
                return true;
            }
        }
        const bool ok = compute(out);
        {
            std::lock_guard<std::mutex> lk(m);
            if (ok) {
                it->packed.assign(out, out + crab::EIG_STRIDE);
                it->ready = true;
            } else {
                it->failed = true;
                auto r = map.equal_range(h);
                for (auto i = r.first; i != r.second; ++i)
                    if (i->second == it) {
                        map.erase(i);
                        break;
                    }
            }
        }
        cv.notify_all();
        return ok;
    }
    void evict_locked()   // drop the least recently used quarter (finished entries only)
    {
        std::vector<std::pair<uint64_t, uint64_t>> old;   // (stamp, hash)
        for (auto &kv : map)
            if (kv.second->ready) old.emplace_back(kv.second->stamp, kv.first);
        std::sort(old.begin(), old.end());
        const size_t drop = old.size() / 4;
        for (size_t i = 0; i < drop; ++i) {
            auto r = map.equal_range(old[i].second);
            for (auto it = r.first; it != r.second; ++it)
                if (it->second->ready && it->second->stamp == old[i].first) {
                    map.erase(it);
                    break;
                }
        }
    }
};

// One persistent host thread per shard of a multi-device handle (shard 0 runs on the caller's thread).
class ShardWorkers {
    struct Slot {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool has_job = false, done = true, stop = false;
    };
    std::vector<std::unique_ptr<Slot>> slots_;

  public:
    void start(int n)
    {
        for (int i = 0; i < n; ++i) {
            slots_.emplace_back(new Slot());
            Slot *s = slots_.back().get();
            s->th = std::thread([s] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(s->m);
                        s->cv.wait(lk, [&] { return s->has_job || s->stop; });
                        if (s->stop) return;
                        job.swap(s->job);
                        s->has_job = false;
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(s->m);
                        s->done = true;
                    }
                    s->cv.notify_all();
                }
            });
        }
    }
    void post(int i, std::function<void()> job)
    {
        Slot *s = slots_[i].get();
        {
            std::lock_guard<std::mutex> lk(s->m);
            s->job = std::move(job);
            s->has_job = true;
            s->done = false;
        }
        s->cv.notify_all();
    }
    void wait(int i)
    {
        Slot *s = slots_[i].get();
        std::unique_lock<std::mutex> lk(s->m);
        s->cv.wait(lk, [&] { return s->done; });
    }
    ~ShardWorkers()
    {
        for (auto &sp : slots_) {
            {
                std::lock_guard<std::mutex> lk(sp->m);
                sp->stop = true;
            }
            sp->cv.notify_all();
            if (sp->th.joinable()) sp->th.join();
        }
    }
};

// The tree compiled for the device: topology tables plus the flat post-order op list.
struct TreeProgram {
    int n_nodes = 0, root = -1, n_leaf_rows = 0;
    std::vector<int> parent, child_start, child, leaf_row;
    std::vector<char> in_leaf_set, reachable;
    std::vector<TreeOp> ops;
    int max_depth = 0, first_gemm_op = -1, first_leaf_row = -1, first_leaf_op = -1, second_leaf_row = -1, third_leaf_row = -1;
    int second_leaf_op = -1, fourth_leaf_row = -1;
    int64_t n_gemm_edges = 0, n_leaf_edges = 0;
    // for path programs (partial updates): the ops that multiply a term into each node, in program order, and the op
    // that crosses the branch above each internal node
    std::vector<std::vector<int>> term_ops;
    std::vector<int> gemm_op_of;
    int n_save_slots = 0;
    // for tree parts: the ops of the subtree below each internal node are the slice [op_begin, op_fin] of the program
    std::vector<int> op_begin, op_fin;
};

// What the kernel needs to know about one program besides its ops.
struct ProgInfo {
    int n_ops = 0, first_gemm_op = -1, first_leaf_row = -1, first_leaf_op = -1, second_leaf_row = -1, third_leaf_row = -1;
    int second_leaf_op = -1, fourth_leaf_row = -1;   // a program that opens with two leaf edges of one node: see finalise_program
    int n_gemm = 0, n_leaf = 0;
    bool cyclic = false;   // the program may run over consecutive tiles in one work item (F_NEXT_TILE, ROW_NEXT_TILE)
};
// A path program: what changes when only the branch above `node` moves (see build_path_program).
struct PathProgram {
    std::vector<TreeOp> ops;
    ProgInfo info;
};
// A tree cut into subtree programs plus a root program (build_tree_parts).
struct TreeParts {
    int target = 0;                       // the part size (products) these cuts were made for
    std::vector<int> cut;                 // cut nodes; part i saves into slot i
    std::vector<PathProgram> parts;       // subtree programs
    PathProgram root;                     // the rest of the tree
};

// Clade tables (subtree site repeats; build_clade_tables).  A clade whose leaves show few distinct sub-patterns is walked
// once per sub-pattern; the message of its root's branch goes into a table that the main program reads like a leaf edge.
struct CladeTables {
    bool built = false;              // for the engine's current tree and alignment (an empty set is a valid outcome)
    struct Clade {
        int root = -1;               // the clade's root node
        int parent = -1;             // the clade whose program gathers from this one's table, -1: the main program
        int level = 1;               // 1: no clade inside; a clade's program runs after those of all lower levels
        int row = 0;                 // its wide row (absolute): alignment row aln_rows + row in its consumer's program
        int64_t U = 0, tiles = 0;    // distinct sub-patterns, and ceil(U / tile width)
        int slot_off = 0;            // first 64 KiB slot of its table inside one class's run of table slots
        PathProgram prog;            // its subtree with inner clades as table leaves, the product on its root's branch, the store
    };
    std::vector<Clade> clades;       // by (level, program order)
    int n_levels = 0;
    std::vector<int> level_begin;    // clades of level l + 1 are [level_begin[l], level_begin[l + 1])
    std::vector<int64_t> row_begin;  // wide rows of launch l (levels first, the main program last) are [row_begin[l], row_begin[l + 1])
    int slots_per_class = 0;
    PathProgram main;                // the tree with every outermost clade replaced by a table leaf
};

struct crab_engine : TreeProgram {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::string err;

    // ---- tree ----
    bool have_tree = false;
    std::vector<DevOp> dev_ops;  // [K][n_ops_pad] programs as last uploaded
    std::vector<int32_t> dev_pidx;  // the (class, node) -> matrix table they were resolved with
    std::vector<int32_t> dev_vec_node;  // ... and which vectors of the batch run path programs (node moved, or -1)
    int dev_n_first = 0, dev_n_second = 0;   // programs in the first launch (whole walks, or tree parts) and in the second
                                             // (path programs, or root programs); 0: one launch
    int64_t dev_bounds_first[9] = {0}, dev_bounds_second[9] = {0};   // per-XCD chunks of the two launches' work items
    int dev_part_target = 0;            // the tree-part size the uploaded programs were cut for (0: uncut)
    bool dev_clades = false;            // the uploaded programs include clade programs + main programs
    std::vector<int32_t> dev_tabled;    // ... for these classes
    int dev_first_off = 0;              // entries of the program list that precede the first launch's (the clade launches')
    struct CladeLaunch {                // one launch of clade programs (one level)
        int list_off = 0, n_entries = 0;      // its entries of the program list
        int64_t items_off = 0, n_items = 0;   // its work items
        int grid = 0;                          // its workgroups
        bool merged = false;                   // all levels in this launch: items wait for the tables they gather from
        int64_t bounds[9] = {0};
    };
    std::vector<CladeLaunch> dev_clade_launches;
    int64_t dev_gemm_extra = 0;         // products of the uploaded clade programs, all sub-patterns and classes (not per pattern)
    CladeTables clades;
    bool clade_tables_on = true;        // crab_set_clade_tables
    int last_n_clades = 0, last_n_launches = 1;   // launches of the pruning kernel in the last evaluation
    std::vector<uint8_t> h_codes;       // host copy of d_codes: [aln_rows][n_pat_pad]
    TreeParts tree_parts;
    int64_t dev_gemm_per_pat = 0, dev_leaf_per_pat = 0;
    std::unordered_map<int, PathProgram> path_programs;   // by moved node, built on first use

    // ---- alignment ----
    bool have_aln = false;
    int n_states = 0, aln_rows = 0;
    int64_t n_sites = 0, n_pat_total = 0, pat_begin = 0, pat_end = 0, n_pat = 0, n_pat_pad = 0;
    std::vector<int64_t> site_to_pat;  // global pattern id per site
    std::vector<double> weights;       // per local pattern

    // ---- device state ----
    DevBuf<DevOp> d_ops;
    DevBuf<uint8_t> d_codes;
    DevBuf<double> d_weights, d_small, d_P, d_stack, d_f, d_patlnl, d_partials, d_lnL, d_byclass;
    DevBuf<int32_t> d_stack_e, d_e, d_beb_cols, d_cls, d_work_ctr, d_deps;
    int dev_n_prog = 0;
    int32_t dev_clade_tiles = 0;        // tiles a merged clade launch announces in all
    bool merged_clade_launch_off = false;   // a merged clade launch of this engine left tables incomplete once: level launches since
    DevBuf<ProgHdr> d_items;
    DevBuf<uint32_t> d_idx32;
    DevBuf<uint8_t> d_codes_alt;
    DevBuf<ProgHdr> d_hdr;
    DevBuf<char> d_save;
    int last_n_path = 0, last_n_parts = 0;
    DevBuf<double> d_beb_F, d_beb_props, d_beb_lnfx, d_beb_w, d_beb_post;
    PinBuf<double> h_eig, h_small, h_out, h_S;
    DevBuf<double> d_S;
    DevBuf<int32_t> d_eig_status;   // kernel 0's convergence flag
    PinBuf<int32_t> h_eig_status;
    PinBuf<int32_t> h_sync_flag;    // set by the device when the tables of a merged clade launch turn out incomplete
    bool device_eigen = false;   // decompose on the device (kernel 0) instead of on the host: crab_set_eigensolver

    EigPool eig_pool;
    int64_t eig_hits = 0, eig_misses = 0;

    // ---- multi-device handle (crab_create_multi): this engine owns no device itself, its shards do ----
    std::vector<crab_engine *> shards;
    std::unique_ptr<ShardWorkers> workers;
    std::unique_ptr<SharedEigCache> shared_eig_owner;
    SharedEigCache *shared_eig = nullptr;   // set on the shards of a multi-device handle
    bool is_multi() const { return !shards.empty(); }

    // ---- last-eval facts ----
    std::vector<int32_t> last_pidx;  // [K][n_nodes] of the last evaluation
    int last_K = 0, last_M = 0;
    double last_device_ms = 0, last_prune_ms = 0, last_eigen_ms = 0;
    int64_t last_dense = 0, last_gather = 0;

    ~crab_engine()
    {
        workers.reset();
        for (crab_engine *sh : shards) delete sh;
        if (is_multi()) return;   // owns no device state of its own
        (void)hipSetDevice(device);
        eig_pool.release();
        d_ops.release(); d_codes.release(); d_weights.release(); d_small.release(); d_P.release();
        d_stack.release(); d_f.release(); d_patlnl.release(); d_partials.release();
        d_lnL.release(); d_byclass.release(); d_stack_e.release(); d_e.release(); d_cls.release(); d_hdr.release(); d_save.release(); d_items.release(); d_work_ctr.release(); d_deps.release(); d_idx32.release(); d_codes_alt.release();
        d_beb_cols.release(); d_beb_F.release(); d_beb_props.release(); d_beb_lnfx.release(); d_beb_w.release(); d_beb_post.release();
        h_eig.release(); h_small.release(); h_out.release(); h_S.release(); d_S.release(); d_eig_status.release(); h_eig_status.release(); h_sync_flag.release();
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (ev_k0) (void)hipEventDestroy(ev_k0);
        if (ev_k1) (void)hipEventDestroy(ev_k1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};


namespace crab_host {


// ---------------------------------------------------------------------------------------------
// error plumbing: run a body, translate exceptions into status codes
// ---------------------------------------------------------------------------------------------
template <typename F>
int guarded(crab_engine *e, F &&body)
{
    try {
        body();
        return CRAB_OK;
    } catch (const ArgError &a) {
        (e ? e->err : g_last_error) = a.msg;
        return a.code;
    } catch (const HipError &h) {
        (e ? e->err : g_last_error) = h.msg;
        return CRAB_ERR_DEVICE;
    } catch (const std::bad_alloc &) {
        (e ? e->err : g_last_error) = "host allocation failed";
        return CRAB_ERR_DEVICE;
    } catch (const std::exception &x) {
        (e ? e->err : g_last_error) = x.what();
        return CRAB_ERR_ARG;
    }
}

[[noreturn]] inline void fail(int code, const std::string &msg) { throw ArgError{code, msg}; }
uint64_t mix64(uint64_t h, uint64_t v);

// ---------------------------------------------------------------------------------------------
// one evaluation
// ---------------------------------------------------------------------------------------------
struct EvalPlan {
    int n_eig = 0;                      // eigensystems to decompose
    const double *Q = nullptr;          // [n_eig][n][n], or
    std::vector<std::pair<double, double>> kw;  // [n_eig] distinct (kappa, omega): unnormalised codon matrices, built on a miss
    int M = 0;                          // transition matrices to build
    std::vector<int32_t> eig_id;        // [M]
    std::vector<double> t_eff;          // [M]
    int K = 0;                          // classes walked by the pruning kernel (all groups)
    int n_groups = 1;                   // parameter vectors evaluated together; K / n_groups classes each
    std::vector<int32_t> pidx;          // [K][n_nodes]
    std::vector<double> props;          // [K]
    int mix_mode = MIX_LOGSUMEXP_CLAMP;
    bool by_class = false;
    bool clamp = true;
    struct Beb {                        // grid integration of site posteriors after the pruning pass (kernel 4)
        enum Mode { WHOLE, PARTIAL, GIVEN };   // whole alignment here; or one pattern slice: its lnfX only / posteriors from the summed lnfX
        Mode mode = WHOLE;
        int G = 0, n_mix = 0;
        const int32_t *cols = nullptr;  // [G][n_mix] class index of each mixture component
        const double *props = nullptr;  // [G][n_mix]
        double *post = nullptr;         // out [n_sites][n_mix]
        double *lnfx = nullptr;         // out [G] or null (in, with GIVEN)
        double *ln_fx = nullptr;        // out: ln f(X) = logsumexp_g lnfx[g]
    };
    const Beb *beb = nullptr;
    const double *post_weights = nullptr;   // [n_groups]: weighted class posteriors instead of a likelihood (kernel 4b)
    double *post_out = nullptr;             // out [n_sites][K / n_groups]
};


// ---- crab_programs.cpp ----
void finalise_program(std::vector<TreeOp> &ops, ProgInfo &info, bool cyclic = false);
void build_tree(TreeProgram *e, int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_ids, int n_leaves);
void build_path_program(const TreeProgram *e, int node, PathProgram &out);
void build_tree_parts(const TreeProgram *e, int target, TreeParts &tp);
void plan_clade_tables(const TreeProgram *e, const uint8_t *codes, int aln_rows, int64_t n_pat, int64_t npp, int n_states,
                       CladeTables &ct, std::vector<uint32_t> &idx, std::vector<uint8_t> &alt);
void build_clade_tables(crab_engine *e);
void resolve_program(const std::vector<TreeOp> &ops, const int32_t *pk, bool save, DevOp *out);
void resolve_programs(const TreeProgram *e, int K, const std::vector<int32_t> &pidx, std::vector<DevOp> &out);

// ---- crab_alignment.cpp ----
// The alignment after pattern compression, before slicing: computed once per crab_set_alignment, shared by the shards
// of a multi-device handle.
struct CompressedAlignment {
    int n_leaves = 0, n_states = 0;
    int64_t n_sites = 0;
    std::vector<uint8_t> cols;          // [n_sites][n_leaves] byte image
    std::vector<int64_t> site_to_pat;   // global pattern id per site
    std::vector<int64_t> first_site;    // representative site per global pattern
    std::vector<double> count;          // multiplicity per global pattern
};
void compress_patterns(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress,
                       std::vector<uint8_t> &cols, std::vector<int64_t> &site_to_pat, std::vector<int64_t> &first_site,
                       std::vector<double> &count);
void compress_alignment(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress, CompressedAlignment &ca);
void upload_alignment(crab_engine *e, const CompressedAlignment &ca, int64_t pat_begin, int64_t pat_end);
void load_alignment(crab_engine *e, const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress,
                    int64_t pat_begin, int64_t pat_end);

// ---- crab_workplan.cpp ----
struct WorkSplit {     // how plan_device_work cut the evaluation
    int G = 1, Kg = 0;             // parameter vectors, classes per vector
    int n_path = 0;                // vectors that run path programs (partial updates)
    int n_parts = 0;               // subtree programs per class (tree parts)
    int n_cl = 0, n_tab = 0;       // clades, and classes whose programs gather from clade tables
    int64_t n_tiles = 0;           // pattern tiles of the alignment
    bool use_clades = false;
};
WorkSplit plan_device_work(crab_engine *e, const EvalPlan &pl, const std::vector<int32_t> *pidx, int M);

// ---- crab_engine.cpp ----
bool path_updates_disabled();
bool tree_parts_disabled();
bool dynamic_items_enabled();
bool clade_level_launches_forced();
bool clade_tables_disabled();
size_t path_save_limit_bytes();
void check_ready(crab_engine *e);
void run_plan(crab_engine *e, const EvalPlan &pl, const double *pi, double *lnL, double *site_lnl, double *by_class_out);
void plan_site_classes(crab_engine *e, EvalPlan &pl, int K, const double *Q, const double *t);

// ---------------------------------------------------------------------------------------------
// multi-device handles: one crab_engine that owns several GPUs of one process (crab_create_multi)
// ---------------------------------------------------------------------------------------------
// Site patterns are independent (likelihood.rs:176-195 never mixes alignment rows): shard r holds a tile-aligned slice of
// the compressed pattern list on its own device, with its own stream and host thread, and every transition matrix.
// A call on the handle runs on all shards at once; log-likelihoods are added in shard order on the host (fixed order:
// bit-identical from run to run), per-site outputs are written by the shard that owns the site.
template <typename F>
int fan_out(crab_engine *e, F &&call)   // call(shard, r) -> status; returns the first failure, message copied to e->err
{
    const int N = (int)e->shards.size();
    std::vector<int> rc(N, CRAB_OK);
    for (int r = 1; r < N; ++r) e->workers->post(r - 1, [&, r] { rc[r] = call(e->shards[r], r); });
    rc[0] = call(e->shards[0], 0);
    for (int r = 1; r < N; ++r) e->workers->wait(r - 1);
    for (int r = 0; r < N; ++r)
        if (rc[r] != CRAB_OK) {
            e->err = "device " + std::to_string(e->shards[r]->device) + " (shard " + std::to_string(r) + "): " + e->shards[r]->err;
            return rc[r];
        }
    return CRAB_OK;
}

// B log-likelihoods per shard, added in shard order
template <typename F>
int multi_sum(crab_engine *e, int B, double *lnL, F &&call)   // call(shard, partial lnL [B]) -> status
{
    if (B < 1 || !lnL) {
        e->err = "bad arguments";
        return CRAB_ERR_ARG;
    }
    const int N = (int)e->shards.size();
    std::vector<double> part((size_t)N * B, 0.0);
    const int rc = fan_out(e, [&](crab_engine *sh, int r) { return call(sh, part.data() + (size_t)r * B); });
    if (rc != CRAB_OK) return rc;
    for (int b = 0; b < B; ++b) {
        double total = 0.0;
        for (int r = 0; r < N; ++r) total += part[(size_t)r * B + b];
        lnL[b] = total;
    }
    return CRAB_OK;
}

// ---- crab_models.cpp ----
struct CodonGraph {
    int row_begin[62];
    int col[526];
    bool ts[526], nonsyn[526];
    double rate(int k, double kappa, double omega) const
    {
        double r = 1.0;
        if (ts[k]) r *= kappa;
        if (nonsyn[k]) r *= omega;
        return r;
    }
};
const CodonGraph &codon_graph();
double codon_mean_rate(double kappa, double omega, const double *pi);
}  // namespace crab_host
