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

namespace {

thread_local std::string g_last_error;

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

double now_ms()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

}  // namespace

uint64_t hash_words(const void *p, size_t bytes, uint64_t seed);
void hash_words2(const void *p, size_t bytes, uint64_t seed, uint64_t out[2]);

namespace crab {
// for entry points that live in other source files (crab_noalloc.hip): the message crab_last_error(NULL) returns
void set_global_error(const std::string &msg) { g_last_error = msg; }
}  // namespace crab

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
        h_eig.release(); h_small.release(); h_out.release(); h_S.release(); d_S.release(); d_eig_status.release(); h_eig_status.release();
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (ev_k0) (void)hipEventDestroy(ev_k0);
        if (ev_k1) (void)hipEventDestroy(ev_k1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

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

[[noreturn]] void fail(int code, const std::string &msg) { throw ArgError{code, msg}; }
uint64_t mix64(uint64_t h, uint64_t v);

// ---------------------------------------------------------------------------------------------
// tree -> program
// ---------------------------------------------------------------------------------------------
// Links the products of a program (a GEMM op's arg is the index of the next GEMM op, whose operand image it stages)
// and lays down the leaf-edge prefetch schedule.  Used for the whole-tree program and for path programs alike.
// `cyclic` (clade programs): the program is laid out to run over consecutive pattern tiles in one work item -- beyond its last
// leaf edge the code queue continues with the first leaf edges of the next tile (ROW_NEXT_TILE), and the last product requests
// their columns and stages the first product's operand image while it runs (F_NEXT_TILE).
void finalise_program(std::vector<TreeOp> &ops, ProgInfo &info, bool cyclic = false)
{
    info = ProgInfo{};
    info.n_ops = (int)ops.size();
    for (const TreeOp &op : ops) {
        info.n_gemm += op.kind == OP_GEMM;
        info.n_leaf += op.kind == OP_LEAF;
    }
    // host program: a GEMM op's arg is the index of the next GEMM op (-1 none)
    int next_gemm_op = -1;
    for (int i = (int)ops.size() - 1; i >= 0; --i)
        if (ops[i].kind == OP_GEMM) {
            ops[i].arg = next_gemm_op;
            next_gemm_op = i;
        }
    info.first_gemm_op = next_gemm_op;
    // ---- pass 4: leaf-edge prefetch schedule.  A wave owns one prefetch tile G.  The columns of a leaf edge are
    // requested into G at the first product (GEMM) after the previous leaf edge was consumed, so that the gather
    // latency hides under MFMA work, or at the end of the op just before it when no product lies in between.
    // When the two ops after a product are both leaf edges (a cherry, typically), the second one's columns are
    // requested by the same product into the registers of its dying B operand, which is the landing tile of the
    // ops that follow.  Codes are queued two leaf edges ahead. ----
    std::vector<int> leaf_ops;
    for (int i = 0; i < (int)ops.size(); ++i) {
        TreeOp &op = ops[i];
        op.pre_node = op.pre_node_b = -1;
        op.pre_row2 = op.pre_row3 = -1;
        op.flags &= ~(F_PRE_EARLY | F_PRE_LATE | F_PRE_DYING | F_IN_LANDING);
        if (op.kind == OP_LEAF) leaf_ops.push_back(i);
    }
    const int n_leaf = (int)leaf_ops.size();
    for (TreeOp &op : ops) op.flags &= ~F_NEXT_TILE;
    // (cyclic needs a prefetch tile, a program that opens with a leaf edge and ends with a non-product op after its last product)
    cyclic = cyclic && prune_has_prefetch() && n_leaf > 0 && leaf_ops[0] == 0 && info.n_gemm > 0 && ops.back().kind != OP_GEMM &&
             ops.back().kind != OP_LEAF;
    auto leaf_row = [&](int k) {
        if (k < n_leaf) return ops[leaf_ops[k]].arg;
        return cyclic && k - n_leaf < n_leaf ? (ops[leaf_ops[k - n_leaf]].arg | ROW_NEXT_TILE) : -1;
    };
    info.first_leaf_row = leaf_row(0);
    info.second_leaf_row = leaf_row(1);
    info.third_leaf_row = leaf_row(2);
    info.first_leaf_op = (n_leaf > 0 && leaf_ops[0] == 0) ? 0 : -1;
    if (!prune_has_prefetch()) return;
    const int reg_slots = prune_reg_stack_slots();
    int li = 0;                                  // leaf_ops[li] = next leaf edge not yet consumed
    int req = info.first_leaf_op == 0 ? 1 : 0;     // how many of the upcoming leaf edges have been requested (0..2)
    // A program that opens with two leaf edges of one node (a cherry; most clade programs): the item's prologue requests both
    // at once, the first into G and the second into the landing tile, which nothing uses before the first product.  (With
    // one request the second gather started when the first had landed: a whole round trip with nothing to hide it.)
    if (req == 1 && n_leaf >= 2 && leaf_ops[1] == 1 && (ops[1].flags & F_MUL_PROD)) {
        ops[1].flags |= F_IN_LANDING;
        info.second_leaf_op = 1;
        info.fourth_leaf_row = leaf_row(3);
        req = 2;
    }
    for (int i = 0; i < (int)ops.size(); ++i) {
        TreeOp &op = ops[i];
        const bool next_is_leaf = i + 1 < (int)ops.size() && ops[i + 1].kind == OP_LEAF;
        auto request_g = [&](int flag) {   // leaf li + req -> G (req must be 0)
            op.flags |= flag;
            op.pre_node = ops[leaf_ops[li]].node;
            op.pre_row2 = leaf_row(li + 2);
            req = 1;
        };
        if (op.kind == OP_LEAF) {
            --req;
            ++li;
            if (req == 0 && next_is_leaf) request_g(F_PRE_LATE);
        } else if (op.kind == OP_GEMM) {
            const int head = li + req;  // leaf edge at the head of the code queue before this op
            if (req == 0 && li < n_leaf) request_g(F_PRE_EARLY);
            const bool global_pop = (op.flags & F_POP_MUL) && ((op.flags & F_POP_SAVED) || op.slot >= reg_slots);
            if (req == 1 && next_is_leaf && li + 1 < n_leaf && leaf_ops[li + 1] == i + 2 && (ops[i + 2].flags & F_MUL_PROD) &&
                !global_pop) {
                op.flags |= F_PRE_DYING;
                op.pre_node_b = ops[i + 2].node;
                ops[i + 2].flags |= F_IN_LANDING;
                req = 2;
            }
            // the requests of this op took 1 or 2 entries off the code queue: refill with the rows that follow
            op.pre_row2 = leaf_row(head + 2);
            op.pre_row3 = leaf_row(head + 3);
        } else if (req == 0 && next_is_leaf) {
            request_g(F_PRE_LATE);
        }
    }
    if (cyclic) {
        int last_gemm = -1;
        for (int i = 0; i < (int)ops.size(); ++i)
            if (ops[i].kind == OP_GEMM) last_gemm = i;
        // no leaf edge may follow the last product (its columns would still be queued when the tail wants the queue's head)
        for (int i = last_gemm + 1; i < (int)ops.size(); ++i) cyclic = cyclic && ops[i].kind != OP_LEAF;
        if (cyclic) {
            // the last product asks for the next tile's head while it runs: the first leaf edge into the prefetch tile (free: no
            // leaf edge follows), the second into its dying operand -- the landing tile of the tile that follows
            TreeOp &lg = ops[last_gemm];
            cyclic = !(lg.flags & (F_PRE_EARLY | F_PRE_DYING | F_POP_MUL | F_PUSH));
            if (cyclic) {
                lg.flags |= F_NEXT_TILE | F_PRE_EARLY;
                lg.arg = info.first_gemm_op;
                lg.pre_node = ops[leaf_ops[0]].node;
                lg.pre_row2 = lg.pre_row3 = -1;
                if (info.second_leaf_op == 1) {
                    lg.flags |= F_PRE_DYING;
                    lg.pre_node_b = ops[1].node;
                }
                info.cyclic = true;
            }
        }
    }
    if (!info.cyclic && n_leaf > 0) {   // (rows of a next tile were handed out above: take them back)
        for (TreeOp &op : ops) {
            if (op.pre_row2 >= 0 && (op.pre_row2 & ROW_NEXT_TILE)) op.pre_row2 = -1;
            if (op.pre_row3 >= 0 && (op.pre_row3 & ROW_NEXT_TILE)) op.pre_row3 = -1;
        }
        if (info.second_leaf_row >= 0 && (info.second_leaf_row & ROW_NEXT_TILE)) info.second_leaf_row = -1;
        if (info.third_leaf_row >= 0 && (info.third_leaf_row & ROW_NEXT_TILE)) info.third_leaf_row = -1;
        if (info.fourth_leaf_row >= 0 && (info.fourth_leaf_row & ROW_NEXT_TILE)) info.fourth_leaf_row = -1;
    }
}

void build_tree(TreeProgram *e, int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_ids,
                int n_leaves)
{
    if (n_nodes < 1 || !parents || (n_leaves > 0 && !leaf_ids) || n_leaves < 0) fail(CRAB_ERR_ARG, "crab_set_tree: bad arguments");
    e->n_nodes = n_nodes;
    e->parent.assign(n_nodes, -1);
    std::vector<char> seen(n_nodes, 0);
    for (int k = 0; k < n_nodes; ++k) {
        const int v = node_ids ? node_ids[k] : k;
        if (v < 0 || v >= n_nodes) fail(CRAB_ERR_ARG, "tree_structure: node id " + std::to_string(v) + " out of range");
        if (parents[k] >= n_nodes) fail(CRAB_ERR_ARG, "tree_structure: parent id " + std::to_string(parents[k]) + " out of range");
        if (seen[v]) fail(CRAB_ERR_ARG, "tree_structure: node id " + std::to_string(v) + " appears more than once");
        e->parent[v] = parents[k] < 0 ? -1 : parents[k];
        seen[v] = 1;   // n_nodes distinct ids in [0, n_nodes): every node is listed exactly once
    }
    // children in order of appearance (likelihood.rs:49-54)
    e->child_start.assign(n_nodes + 1, 0);
    for (int k = 0; k < n_nodes; ++k)
        if (parents[k] >= 0) e->child_start[parents[k] + 1]++;
    for (int v = 0; v < n_nodes; ++v) e->child_start[v + 1] += e->child_start[v];
    e->child.assign(std::max(1, e->child_start[n_nodes]), -1);
    {
        std::vector<int> fill(n_nodes, 0);
        for (int k = 0; k < n_nodes; ++k)
            if (parents[k] >= 0) e->child[e->child_start[parents[k]] + fill[parents[k]]++] = node_ids ? node_ids[k] : k;
    }
    e->root = -1;
    for (int v = 0; v < n_nodes; ++v)
        if (e->parent[v] < 0) {
            e->root = v;
            break;
        }
    if (e->root < 0) fail(CRAB_ERR_NO_ROOT, "No root node found");
    e->in_leaf_set.assign(n_nodes, 0);
    e->leaf_row.assign(n_nodes, -1);
    e->n_leaf_rows = n_leaves;
    for (int r = 0; r < n_leaves; ++r) {
        const int v = leaf_ids[r];
        if (v < 0 || v >= n_nodes) fail(CRAB_ERR_ARG, "leaf_node_ids: node id " + std::to_string(v) + " out of range");
        e->in_leaf_set[v] = 1;
        e->leaf_row[v] = r;
    }
    if (e->in_leaf_set[e->root]) fail(CRAB_ERR_ARG, "the root node is listed as a leaf: nothing to prune");

    // ---- pass 1: iterative post-order over nodes reachable from the root; cycle guard ----
    std::vector<int> order;
    order.reserve(n_nodes);
    e->reachable.assign(n_nodes, 0);
    {
        std::vector<std::pair<int, int>> st;
        st.emplace_back(e->root, 0);
        e->reachable[e->root] = 1;
        while (!st.empty()) {
            auto &[v, k] = st.back();
            const bool expand = !e->in_leaf_set[v];
            if (expand && e->child_start[v] + k < e->child_start[v + 1]) {
                const int c = e->child[e->child_start[v] + k];
                ++k;
                if (e->reachable[c]) fail(CRAB_ERR_ARG, "tree_structure contains a cycle or a repeated node");
                e->reachable[c] = 1;
                st.emplace_back(c, 0);
            } else {
                order.push_back(v);
                st.pop_back();
            }
        }
    }
    // ---- pass 2: stack need per internal node (Sethi-Ullman) ----
    auto is_internal = [&](int v) { return !e->in_leaf_set[v]; };
    std::vector<int> need(n_nodes, 0), height(n_nodes, 0);  // height: internal levels below the node
    std::vector<std::vector<int>> int_children(n_nodes);
    for (int v : order) {
        if (!is_internal(v)) continue;
        auto &ic = int_children[v];
        for (int ci = e->child_start[v]; ci < e->child_start[v + 1]; ++ci) {
            const int c = e->child[ci];
            if (is_internal(c)) ic.push_back(c);
        }
        std::stable_sort(ic.begin(), ic.end(), [&](int x, int y) { return need[x] > need[y]; });
        int nd = 0, h = 0;
        for (size_t i = 0; i < ic.size(); ++i) {
            nd = std::max(nd, need[ic[i]] + (i > 0 ? 1 : 0));
            h = std::max(h, height[ic[i]] + 1);
        }
        need[v] = nd;
        height[v] = h;
    }
    e->max_depth = need[e->root];

    // ---- pass 3: emit ops, iteratively ----
    e->ops.clear();
    e->n_gemm_edges = e->n_leaf_edges = 0;
    struct Frame {
        int v, depth;
        size_t next_child;  // index into int_children[v]
        bool any_term;
        int n_terms;        // terms multiplied into this node so far (a rescale is forced every 4th)
    };
    auto emit = [&](int kind, int node, int flags, int arg, int slot) {
        TreeOp op{};
        op.kind = kind;
        op.node = node;
        op.flags = flags;
        op.arg = arg;
        op.slot = slot;
        e->ops.push_back(op);
    };
    std::vector<Frame> fs;
    e->op_begin.assign(n_nodes, -1);
    e->op_fin.assign(n_nodes, -1);
    e->op_begin[e->root] = 0;
    fs.push_back({e->root, 0, 0, false, 0});
    while (!fs.empty()) {
        Frame &f = fs.back();
        const auto &ic = int_children[f.v];
        if (f.next_child < ic.size()) {
            const size_t i = f.next_child++;
            if (i > 0) e->ops.back().flags |= F_PUSH, e->ops.back().slot = f.depth;  // save the running product
            e->op_begin[ic[i]] = (int)e->ops.size();
            fs.push_back({ic[i], f.depth + (i > 0 ? 1 : 0), 0, false, 0});
            continue;
        }
        // all internal-child subtrees of f.v have been emitted *except* that each child's GEMM must follow its
        // subtree immediately; that is done when the child frame is popped (below).  Here: leaf children, FIN.
        const int v = f.v;
        bool any = f.any_term;
        for (int ci = e->child_start[v]; ci < e->child_start[v + 1]; ++ci) {
            const int c = e->child[ci];
            if (is_internal(c)) continue;
            int flags = any ? F_MUL_PROD : 0;
            if (++f.n_terms % 4 == 0) flags |= F_RESCALE;
            emit(OP_LEAF, c, flags, e->leaf_row[c], 0);
            e->n_leaf_edges++;
            any = true;
        }
        if (!any) emit(OP_CONST, v, 0, 1, 0);  // childless non-leaf node: all ones (likelihood.rs:178)
        // A finished node is normalised (power-of-two rescaling, exponent booked) on every second level and at the
        // root: between two normalisations a conditional likelihood passes through at most two products and a
        // handful of sibling multiplications, far from the fp64 exponent range even on tiny branches.
        e->ops.back().flags |= F_FIN | ((height[v] % 2 == 0 || v == e->root) ? F_RESCALE : 0);
        e->op_fin[v] = (int)e->ops.size() - 1;
        const int depth = f.depth;
        fs.pop_back();
        if (!fs.empty()) {
            // v is an internal child of the frame below: cross its branch now
            Frame &p = fs.back();
            const bool first = !p.any_term;
            int flags = 0, slot = 0;
            if (!first) {
                flags |= F_POP_MUL;
                slot = p.depth;
            }
            (void)depth;
            if (++p.n_terms % 4 == 0) flags |= F_RESCALE;
            emit(OP_GEMM, v, flags, -1, slot);
            e->n_gemm_edges++;
            p.any_term = true;
        }
    }
    // Stack slots are numbered by frame depth, and the inner frames see most of the traffic (100 taxa: 5 / 17 / 9 saves
    // at depths 0 / 1 / 2; 1000 taxa: 15 / 51 / 110 / 110 / 39).  The kernel keeps the lowest-numbered slots in
    // registers and the rest in global memory, so renumber the slots by how often they are written, busiest first.
    {
        std::vector<int64_t> writes(std::max(1, e->max_depth) + 1, 0);
        for (const TreeOp &op : e->ops)
            if ((op.flags & F_PUSH) && op.slot >= 0 && op.slot < (int)writes.size()) ++writes[op.slot];
        std::vector<int> order(writes.size()), rank(writes.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return writes[a] > writes[b]; });
        for (size_t r = 0; r < order.size(); ++r) rank[order[r]] = (int)r;
        for (TreeOp &op : e->ops)
            if ((op.flags & (F_PUSH | F_POP_MUL)) && op.slot >= 0 && op.slot < (int)rank.size()) op.slot = rank[op.slot];
    }
    // save slots of the partial-update path: three per internal edge (crab_kernels.h, "save area")
    e->term_ops.assign(n_nodes, {});
    e->gemm_op_of.assign(n_nodes, -1);
    e->n_save_slots = 0;
    for (int i = 0; i < (int)e->ops.size(); ++i) {
        TreeOp &op = e->ops[i];
        op.sv = 0;
        if (op.kind == OP_GEMM) {
            op.sv = e->n_save_slots;
            e->n_save_slots += 3;
            e->gemm_op_of[op.node] = i;
        }
        if (op.kind == OP_GEMM || op.kind == OP_LEAF) e->term_ops[e->parent[op.node]].push_back(i);
    }
    ProgInfo info;
    finalise_program(e->ops, info);
    e->first_gemm_op = info.first_gemm_op;
    e->first_leaf_row = info.first_leaf_row;
    e->first_leaf_op = info.first_leaf_op;
    e->second_leaf_row = info.second_leaf_row;
    e->third_leaf_row = info.third_leaf_row;
    e->second_leaf_op = info.second_leaf_op;
    e->fourth_leaf_row = info.fourth_leaf_row;
}

// The program that recomputes the root's conditional likelihood when only the branch above `node` has changed, on the
// tiles the base vector's walk saved (crab_kernels.h, "save area"): it starts from the saved operand of the moved edge
// (or, for a leaf edge, from the parent's running product over its internal children) and climbs to the root; at each
// node on the way the terms of the other children come from the save area (internal children) or are gathered again
// (leaf children), multiplied in the order, with the rescaling points, of the whole-tree program -- so the result
// carries the bits a whole walk of the perturbed vector would produce.
void build_path_program(const TreeProgram *e, int node, PathProgram &out)
{
    const std::vector<TreeOp> &ops = e->ops;
    auto plain = [&](int i) {   // the op as the whole-tree program has it, minus stack traffic and prefetch schedule
        TreeOp o = ops[i];
        o.flags &= ~(F_PUSH | F_PRE_EARLY | F_PRE_LATE | F_PRE_DYING | F_IN_LANDING);
        o.pre_node = o.pre_node_b = o.pre_row2 = o.pre_row3 = -1;
        return o;
    };
    auto load = [&](int slot, int flags) {
        TreeOp o{};
        o.kind = OP_LOAD;
        o.node = -1;
        o.flags = flags;
        o.arg = 0;
        o.slot = 0;
        o.pre_node = o.pre_node_b = o.pre_row2 = o.pre_row3 = -1;
        o.sv = slot;
        return o;
    };
    out.ops.clear();
    int cur;
    if (e->in_leaf_set[node]) {
        const int p = e->parent[node];
        const std::vector<int> &terms = e->term_ops[p];
        int last_gemm = -1;
        for (int t : terms)
            if (ops[t].kind == OP_GEMM) last_gemm = t;
        if (last_gemm >= 0) out.ops.push_back(load(ops[last_gemm].sv + 1, 0));   // running product over the internal children
        for (int t : terms)
            if (ops[t].kind == OP_LEAF) out.ops.push_back(plain(t));
        cur = p;
    } else {
        out.ops.push_back(load(ops[e->gemm_op_of[node]].sv, 0));                   // L_node, the operand of the moved edge
        cur = node;
    }
    while (cur != e->root) {
        const int p = e->parent[cur];
        const std::vector<int> &terms = e->term_ops[p];
        size_t j = 0;
        while (j < terms.size() && terms[j] != e->gemm_op_of[cur]) ++j;
        TreeOp g = plain(terms[j]);
        if (g.flags & F_POP_MUL) {      // the running product before this child is what the preceding sibling's op left
            g.flags |= F_POP_SAVED;
            g.slot = ops[terms[j - 1]].sv + 1;
        }
        out.ops.push_back(g);
        for (size_t i = j + 1; i < terms.size(); ++i) {
            const TreeOp &o = ops[terms[i]];
            if (o.kind == OP_GEMM)      // a later internal sibling: its message, multiplied in where its product was
                out.ops.push_back(load(o.sv + 2, F_MUL_PROD | (o.flags & (F_RESCALE | F_FIN))));
            else
                out.ops.push_back(plain(terms[i]));
        }
        cur = p;
    }
    finalise_program(out.ops, out.info);
}

// Tree parts (a finer work quantum).  A work item of the pruning kernel is a whole-tree walk by one workgroup; when an
// evaluation has few items per resident workgroup (one GPU's share of a sharded alignment) whole CUs idle or run one
// workgroup where two would fit.  The tree is then cut below a set of internal nodes: each subtree becomes a program of
// its own that ends by saving its conditional likelihoods (launch 1: parts x classes x pattern tiles items), and the
// root program loads them where the subtrees were (launch 2).  The ops, their order and the rescaling points are those
// of the whole program, so the result is bit-identical.
void build_tree_parts(const TreeProgram *e, int target, TreeParts &tp)
{
    const std::vector<TreeOp> &ops = e->ops;
    tp = TreeParts{};
    tp.target = target;
    auto work = [&](int v) {   // products inside the subtree below v
        int n = 0;
        for (int i = e->op_begin[v]; i <= e->op_fin[v]; ++i) n += ops[i].kind == OP_GEMM;
        return n;
    };
    // top-down: a subtree larger than the target gives way to the subtrees of its internal children
    std::vector<int> frontier{e->root};
    while (!frontier.empty()) {
        const int v = frontier.back();
        frontier.pop_back();
        const int w = work(v);
        if (v != e->root && w <= target) {
            if (w >= 2) tp.cut.push_back(v);     // smaller ones stay in the root program
            continue;
        }
        for (int t : e->term_ops[v])
            if (ops[t].kind == OP_GEMM) frontier.push_back(ops[t].node);
    }
    std::sort(tp.cut.begin(), tp.cut.end(), [&](int a, int b) { return e->op_begin[a] < e->op_begin[b]; });
    auto plain = [&](int i) {
        TreeOp o = ops[i];
        o.flags &= ~(F_PRE_EARLY | F_PRE_LATE | F_PRE_DYING | F_IN_LANDING);
        o.pre_node = o.pre_node_b = o.pre_row2 = o.pre_row3 = -1;
        return o;
    };
    auto transfer = [&](int slot, int flags) {
        TreeOp o{};
        o.kind = OP_LOAD;
        o.node = -1;
        o.flags = flags;
        o.pre_node = o.pre_node_b = o.pre_row2 = o.pre_row3 = -1;
        o.sv = slot;
        return o;
    };
    tp.parts.resize(tp.cut.size());
    size_t next = 0;
    for (int i = 0; i < (int)ops.size();) {
        if (next < tp.cut.size() && i == e->op_begin[tp.cut[next]]) {
            const int c = tp.cut[next];
            PathProgram &pp = tp.parts[next];
            for (int j = e->op_begin[c]; j <= e->op_fin[c]; ++j) pp.ops.push_back(plain(j));
            pp.ops.push_back(transfer((int)next, F_SAVE));
            finalise_program(pp.ops, pp.info);
            tp.root.ops.push_back(transfer((int)next, 0));      // L_c, where the subtree's ops were
            i = e->op_fin[c] + 1;
            ++next;
        } else {
            tp.root.ops.push_back(plain(i));
            ++i;
        }
    }
    finalise_program(tp.root.ops, tp.root.info);
}

// Clade tables (subtree site repeats).  On alignments of related sequences the leaves of a small clade show far fewer
// distinct sub-patterns than the alignment has patterns (configs[1]: a 6-leaf clade ~15 % of 9 953), and everything the
// pruning pass computes below the clade's root -- its conditional likelihoods and the message of the branch above it --
// depends on the sub-pattern only.  Chosen clades are therefore walked once per distinct sub-pattern by programs of their
// own (the clade's subtree ops, the product on its root's branch, a store into the clade's table), and whoever contains the
// clade -- a larger clade's program, or the main program -- replaces subtree and product by one leaf-style gather from that
// table, indexed per (sub-)pattern.  Clades nest: level 1 holds none, a clade of level l gathers from clades of lower levels
// and is walked over ITS distinct sub-patterns; one launch per level, the main programs last.
// Same ops, same order, same rescaling points as the whole-tree program; a table entry is the tile the whole walk would
// hold after that product, with its power-of-two exponent folded in (exact), and every later normalisation brings the
// running product back to the same mantissas: results are bit-identical to whole walks (tests/test_clade_tables.py).
// Choice: clades of at most 64 leaves (the folded exponent stays far inside the fp64 range), minimising products by a
// dynamic programme over (node, number of patterns it is walked for): walking a node costs one product per pattern plus
// its internal children in the same context; tabling it costs a gather per pattern plus, once, its product and its
// children over its own padded sub-patterns.
// (host only: `codes` is the byte code table [aln_rows][npp] of the n_pat patterns; idx holds the wide rows of every launch,
// alt one code table per level)
void plan_clade_tables(const TreeProgram *e, const uint8_t *codes, int aln_rows, int64_t n_pat, int64_t npp, int n_states,
                       CladeTables &ct, std::vector<uint32_t> &idx, std::vector<uint8_t> &alt)
{
    ct = CladeTables{};
    ct.built = true;
    idx.clear();
    alt.clear();
    const int nn = e->n_nodes, tile = prune_tile_width();
    if (n_pat < 4 * tile || e->n_gemm_edges < 2) return;
    auto env_num = [](const char *name, double dflt) {
        const char *sv = std::getenv(name);
        return sv && sv[0] ? std::atof(sv) : dflt;
    };
    // tuning knobs (experiments: scripts/gpu_clade_levels.sh, gpu_clade_sweep2.sh); the defaults are the measured optimum over
    // configs[1] and the target workload: deeper nesting executes fewer products but adds launches of ever smaller items
    const int MAX_LEAVES = std::max(2, std::min(64, (int)env_num("CRAB_CLADE_MAX_LEAVES", 64)));
    const int MAX_LEVELS = std::max(1, std::min(8, (int)env_num("CRAB_CLADE_MAX_LEVELS", 3)));
    const double GATHER_COST = env_num("CRAB_CLADE_GATHER_COST", 0.15);     // a table gather + multiply per pattern, in products
    const double PREPASS_COST = env_num("CRAB_CLADE_PREPASS_COST", 1.0);    // a product of a clade program against one of the main program
    const double ITEM_COST = env_num("CRAB_CLADE_ITEM_COST", 1.0);          // fixed cost of a clade work item (prologue), in products x patterns of a tile
    const std::vector<TreeOp> &ops = e->ops;
    auto internal_children = [&](int v, std::vector<int> &out) {
        out.clear();
        for (int ci = e->child_start[v]; ci < e->child_start[v + 1]; ++ci)
            if (!e->in_leaf_set[e->child[ci]]) out.push_back(e->child[ci]);
    };
    // post-order over the internal nodes = order of their last op
    std::vector<int> internal;
    for (int v = 0; v < nn; ++v)
        if (e->reachable[v] && !e->in_leaf_set[v] && e->op_fin[v] >= 0) internal.push_back(v);
    std::sort(internal.begin(), internal.end(), [&](int a, int b) { return e->op_fin[a] < e->op_fin[b]; });
    // the plan keeps a 4-byte sub-pattern id per (candidate node, pattern) on the host while it is made: alignments for which that
    // would pass CRAB_CLADE_PLAN_GB (default 8) are walked whole
    if ((double)internal.size() * (double)n_pat * 4.0 > env_num("CRAB_CLADE_PLAN_GB", 8.0) * 1073741824.0) return;
    // ---- distinct sub-patterns of every candidate node, bottom-up: a node's sub-pattern is the tuple of its children's ----
    std::vector<int> n_rows(nn, 0);
    std::vector<char> cand(nn, 0);
    std::vector<std::vector<int32_t>> ids(nn), reps(nn);     // per candidate: sub-pattern of every pattern; first pattern of every sub-pattern
    size_t cap = 64;
    while (cap < (size_t)n_pat * 2) cap <<= 1;
    std::vector<int32_t> slot_pat(cap);
    std::vector<const int32_t *> key_i;                       // the tuple's int components (children that are candidates) ...
    std::vector<const uint8_t *> key_b;                       // ... and its byte components (leaf children)
    for (int v : internal) {
        bool ok = v != e->root;
        key_i.clear(), key_b.clear();
        for (int ci = e->child_start[v]; ci < e->child_start[v + 1]; ++ci) {
            const int c = e->child[ci];
            if (e->in_leaf_set[c]) {
                n_rows[v] += 1;
                key_b.push_back(codes + (size_t)e->leaf_row[c] * npp);
            } else {
                n_rows[v] += n_rows[c];
                if (!cand[c] || 2 * (int64_t)reps[c].size() > n_pat) ok = false;   // the count only grows towards the root
                else key_i.push_back(ids[c].data());
            }
        }
        if (!ok || n_rows[v] < 2 || n_rows[v] > MAX_LEAVES) continue;
        std::fill(slot_pat.begin(), slot_pat.end(), -1);
        std::vector<int32_t> &id = ids[v], &rp = reps[v];
        id.assign(n_pat, 0);
        for (int64_t p = 0; p < n_pat; ++p) {
            uint64_t h = 0x51ed270b7a3f11c5ull;
            for (const int32_t *k : key_i) h = mix64(h, (uint64_t)(uint32_t)k[p]);
            for (const uint8_t *k : key_b) h = mix64(h, k[p]);
            size_t at = (size_t)h & (cap - 1);
            for (;;) {
                const int32_t q = slot_pat[at];
                if (q < 0) {
                    slot_pat[at] = (int32_t)p;
                    id[p] = (int32_t)rp.size();
                    rp.push_back((int32_t)p);
                    break;
                }
                bool same = true;
                for (size_t i = 0; i < key_i.size() && same; ++i) same = key_i[i][q] == key_i[i][p];
                for (size_t i = 0; i < key_b.size() && same; ++i) same = key_b[i][q] == key_b[i][p];
                if (same) {
                    id[p] = id[q];
                    break;
                }
                at = (at + 1) & (cap - 1);
            }
        }
        cand[v] = 1;
    }
    auto u_pad = [&](int v) { return ((int64_t)reps[v].size() + tile - 1) / tile * tile; };
    // ---- the dynamic programme: cost of the subtree below v (its own branch included) when walked for ctx patterns ----
    struct Choice {
        double cost;
        bool table;
    };
    std::map<std::pair<int, int64_t>, Choice> memo;
    std::vector<double> table_cost(nn, -1.0);
    std::vector<int> depth_left(nn, 0);
    std::function<Choice(int, int64_t, int)> cost = [&](int v, int64_t ctx, int levels_left) -> Choice {
        const auto key = std::make_pair(v, ctx * 16 + levels_left);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        std::vector<int> ic;
        internal_children(v, ic);
        double walk = (double)ctx;
        for (int c : ic) walk += cost(c, ctx, levels_left).cost;
        Choice out{walk, false};
        if (cand[v] && levels_left > 0) {
            const int64_t up = u_pad(v);
            double t = PREPASS_COST * (double)up + ITEM_COST * (double)up;
            for (int c : ic) t += PREPASS_COST * cost(c, up, levels_left - 1).cost;
            t += GATHER_COST * (double)ctx;
            if (t < walk) out = Choice{t, true};
        }
        memo.emplace(key, out);
        return out;
    };
    // ---- extract the chosen clades, outermost first ----
    struct Pick {
        int v, parent;
        int64_t ctx;
        int levels_left;
    };
    std::vector<Pick> todo;
    {
        std::vector<int> ic;
        internal_children(e->root, ic);
        for (int c : ic) todo.push_back({c, -1, n_pat, MAX_LEVELS});
    }
    std::vector<CladeTables::Clade> found;
    while (!todo.empty()) {
        const Pick pk = todo.back();
        todo.pop_back();
        const Choice ch = cost(pk.v, pk.ctx, pk.levels_left);
        std::vector<int> ic;
        internal_children(pk.v, ic);
        if (ch.table) {
            CladeTables::Clade cl;
            cl.root = pk.v;
            cl.parent = pk.parent;
            cl.U = (int64_t)reps[pk.v].size();
            cl.tiles = (cl.U + tile - 1) / tile;
            found.push_back(std::move(cl));
            const int me = (int)found.size() - 1;
            for (int c : ic) todo.push_back({c, me, u_pad(pk.v), pk.levels_left - 1});
        } else {
            for (int c : ic) todo.push_back({c, pk.parent, pk.ctx, pk.levels_left});
        }
    }
    if (found.empty()) return;
    // levels bottom-up (children were found after their parents)
    for (int i = (int)found.size() - 1; i >= 0; --i)
        if (found[i].parent >= 0) found[found[i].parent].level = std::max(found[found[i].parent].level, found[i].level + 1);
    std::vector<int> order(found.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return found[a].level != found[b].level ? found[a].level < found[b].level : e->op_begin[found[a].root] < e->op_begin[found[b].root];
    });
    std::vector<int> new_index(found.size());
    for (size_t i = 0; i < order.size(); ++i) new_index[order[i]] = (int)i;
    ct.clades.resize(found.size());
    for (size_t i = 0; i < order.size(); ++i) {
        ct.clades[i] = std::move(found[order[i]]);
        if (ct.clades[i].parent >= 0) ct.clades[i].parent = new_index[ct.clades[i].parent];
    }
    const int n_cl = (int)ct.clades.size();
    ct.n_levels = ct.clades.back().level;
    ct.level_begin.assign(ct.n_levels + 1, n_cl);
    for (int i = n_cl - 1; i >= 0; --i) ct.level_begin[ct.clades[i].level - 1] = i;
    ct.level_begin[ct.n_levels] = n_cl;
    // wide rows: launch l < n_levels serves the clades of level l + 1, launch n_levels the main program
    auto launch_of_consumer = [&](int parent) { return parent < 0 ? ct.n_levels : ct.clades[parent].level - 1; };
    std::vector<int64_t> rows_in(ct.n_levels + 1, 0);
    for (auto &cl : ct.clades) cl.row = (int)rows_in[launch_of_consumer(cl.parent)]++;
    ct.row_begin.assign(ct.n_levels + 2, 0);
    for (int l = 0; l <= ct.n_levels; ++l) ct.row_begin[l + 1] = ct.row_begin[l] + rows_in[l];
    // (rows are absolute from here on: every launch reads the wide rows through the same base, so that clade programs of
    // several levels can share one launch)
    for (auto &cl : ct.clades) cl.row += (int)ct.row_begin[launch_of_consumer(cl.parent)];
    idx.assign((size_t)ct.row_begin[ct.n_levels + 1] * npp, 0u);
    alt.assign((size_t)ct.n_levels * aln_rows * npp, (uint8_t)n_states);
    auto rows_below = [&](int c) {
        std::vector<int> rr, st{c};
        while (!st.empty()) {
            const int v = st.back();
            st.pop_back();
            if (e->in_leaf_set[v]) {
                rr.push_back(e->leaf_row[v]);
                continue;
            }
            for (int ci = e->child_start[v]; ci < e->child_start[v + 1]; ++ci) st.push_back(e->child[ci]);
        }
        return rr;
    };
    int slot = 0;
    for (int j = 0; j < n_cl; ++j) {
        CladeTables::Clade &cl = ct.clades[j];
        // where its consumer looks it up: per pattern (main program) or per distinct sub-pattern of the consuming clade
        uint32_t *row = idx.data() + (size_t)cl.row * npp;
        if (cl.parent < 0) {
            for (int64_t p = 0; p < n_pat; ++p) row[p] = (uint32_t)ids[cl.root][p];
        } else {
            const std::vector<int32_t> &prep = reps[ct.clades[cl.parent].root];
            for (size_t u = 0; u < prep.size(); ++u) row[u] = (uint32_t)ids[cl.root][prep[u]];
        }
        // the codes its own program reads: its distinct sub-patterns on the rows below it
        uint8_t *lvl = alt.data() + (size_t)(cl.level - 1) * aln_rows * npp;
        const std::vector<int32_t> &rp = reps[cl.root];
        for (int r : rows_below(cl.root))
            for (size_t u = 0; u < rp.size(); ++u) lvl[(size_t)r * npp + u] = codes[(size_t)r * npp + rp[u]];
        // a table is addressed like the gather image of a (pseudo) matrix: it starts half a slot in and runs on
        cl.slot_off = slot;
        slot += (int)((P_IMAGE * sizeof(double) + (size_t)cl.tiles * tile * 512 + P_ELEMS * sizeof(double) - 1) / (P_ELEMS * sizeof(double)));
    }
    ct.slots_per_class = slot;
    // ---- programs: a slice of the whole-tree program with the consumer's own clades replaced by table leaves ----
    auto plain = [&](int i) {
        TreeOp o = ops[i];
        o.flags &= ~(F_PRE_EARLY | F_PRE_LATE | F_PRE_DYING | F_IN_LANDING);
        o.pre_node = o.pre_node_b = o.pre_row2 = o.pre_row3 = -1;
        return o;
    };
    std::vector<int> clade_at(ops.size(), -1);          // op index where a clade's subtree begins -> clade
    for (int j = 0; j < n_cl; ++j) clade_at[e->op_begin[ct.clades[j].root]] = -1;   // (filled per consumer below)
    auto rewrite = [&](int lo, int hi, int consumer, std::vector<TreeOp> &out) {
        for (int j = 0; j < n_cl; ++j)
            if (ct.clades[j].parent == consumer) clade_at[e->op_begin[ct.clades[j].root]] = j;
        for (int i = lo; i <= hi;) {
            const int j = clade_at[i];
            if (j >= 0) {
                // the table stands where the product stood -- multiplied into the parent's running product if the product
                // popped one (then nothing needs the push that saved it), with the product's rescale / push / last-term flags
                const int c = ct.clades[j].root, gi = e->gemm_op_of[c];
                const TreeOp &gw = ops[gi];
                TreeOp leaf{};
                leaf.kind = OP_LEAF;
                leaf.node = c;
                leaf.arg = aln_rows + ct.clades[j].row;
                leaf.flags = (gw.flags & (F_RESCALE | F_FIN | F_PUSH)) | ((gw.flags & F_POP_MUL) ? F_MUL_PROD : 0);
                leaf.slot = gw.slot;
                leaf.pre_node = leaf.pre_node_b = leaf.pre_row2 = leaf.pre_row3 = -1;
                if ((gw.flags & F_POP_MUL) && !out.empty()) out.back().flags &= ~F_PUSH;
                out.push_back(leaf);
                i = gi + 1;
            } else {
                out.push_back(plain(i));
                ++i;
            }
        }
        for (int j = 0; j < n_cl; ++j)
            if (ct.clades[j].parent == consumer) clade_at[e->op_begin[ct.clades[j].root]] = -1;
    };
    for (int j = 0; j < n_cl; ++j) {
        CladeTables::Clade &cl = ct.clades[j];
        rewrite(e->op_begin[cl.root], e->op_fin[cl.root], j, cl.prog.ops);
        TreeOp g = plain(e->gemm_op_of[cl.root]);     // the message of the branch above the clade: P_c L_c, nothing else
        g.flags = 0;
        g.slot = 0;
        cl.prog.ops.push_back(g);
        TreeOp st{};
        st.kind = OP_LOAD;
        st.node = -1;
        st.flags = F_SAVE | F_TABLE;
        st.arg = j;
        st.pre_node = st.pre_node_b = st.pre_row2 = st.pre_row3 = -1;
        cl.prog.ops.push_back(st);
        finalise_program(cl.prog.ops, cl.prog.info, true);
    }
    rewrite(0, (int)ops.size() - 1, -1, ct.main.ops);
    finalise_program(ct.main.ops, ct.main.info);
}

void build_clade_tables(crab_engine *e)
{
    std::vector<uint32_t> idx;
    std::vector<uint8_t> alt;
    if (e->h_codes.size() != (size_t)e->aln_rows * e->n_pat_pad) {
        e->clades = CladeTables{};
        e->clades.built = true;
        return;
    }
    plan_clade_tables(e, e->h_codes.data(), e->aln_rows, e->n_pat, e->n_pat_pad, e->n_states, e->clades, idx, alt);
    if (e->clades.clades.empty()) return;
    HIP_CHECK(hipSetDevice(e->device));
    e->d_idx32.ensure(idx.size());
    e->d_codes_alt.ensure(alt.size());
    HIP_CHECK(hipMemcpyAsync(e->d_idx32.p, idx.data(), idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipMemcpyAsync(e->d_codes_alt.p, alt.data(), alt.size(), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
}

// Packs one program for the device: matrix indices (through pk, the program's node -> matrix table) instead of node
// ids, and on each leaf edge the alignment row of the leaf edge after it (whose codes the kernel requests one edge
// ahead).  `save`: GEMM ops store their tiles in the save area (the base vector of a batch with partial updates).
void resolve_program(const std::vector<TreeOp> &ops, const int32_t *pk, bool save, DevOp *out)
{
    const size_t n_ops = ops.size();
    int32_t row = -1;
    std::vector<int32_t> next_leaf_row(n_ops, -1);
    for (int i = (int)n_ops - 1; i >= 0; --i)
        if (ops[i].kind == OP_LEAF) {
            next_leaf_row[i] = row;
            row = ops[i].arg;
        }
    for (size_t i = 0; i < n_ops; ++i) {
        const TreeOp &op = ops[i];
        DevOp d{};
        const int flags = op.flags | ((save && op.kind == OP_GEMM) ? F_SAVE : 0);
        d.w0 = op.kind | (flags << 2) | (op.slot << (2 + OP_FLAG_BITS));
        d.pre_mat = op.pre_node >= 0 ? pk[op.pre_node] : -1;
        d.pre_mat_b = op.pre_node_b >= 0 ? pk[op.pre_node_b] : -1;
        d.pre_row2 = op.pre_row2;
        d.pre_row3 = op.pre_row3;
        d.mat = (op.kind == OP_CONST || op.kind == OP_LOAD) ? 0 : pk[op.node];
        d.arg = op.kind == OP_GEMM ? (op.arg >= 0 ? pk[ops[op.arg].node] : -1) : op.kind == OP_LEAF ? next_leaf_row[i] : op.arg;
        d.sv = op.sv;
        out[i] = d;
    }
}

// Resolves the topology-only program into K packed device programs, one per class.
void resolve_programs(const TreeProgram *e, int K, const std::vector<int32_t> &pidx, std::vector<DevOp> &out)
{
    const size_t stride = (size_t)dev_ops_padded((int64_t)e->ops.size());
    DevOp filler{};
    filler.w0 = OP_CONST;
    out.assign((size_t)K * stride, filler);
    for (int k = 0; k < K; ++k) resolve_program(e->ops, pidx.data() + (size_t)k * e->n_nodes, false, out.data() + (size_t)k * stride);
}

// ---------------------------------------------------------------------------------------------
// alignment -> compressed, sliced, byte-coded patterns on the device
// ---------------------------------------------------------------------------------------------
uint64_t mix64(uint64_t h, uint64_t v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return h;
}

// Collapses identical alignment columns.  cols: column-major byte image [n_sites][n_leaves] (output),
// site_to_pat[s] = pattern id, first_site[p] = first site showing pattern p, count[p] = multiplicity.
// Pattern ids follow the shard order below (compress != 0; sites as they come otherwise): a function of the alignment alone, so
// the pattern order (and every sum over it) is deterministic and the same on every rank.
void compress_patterns(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress,
                       std::vector<uint8_t> &cols, std::vector<int64_t> &site_to_pat, std::vector<int64_t> &first_site,
                       std::vector<double> &count)
{
    // column-major byte image with missing data canonicalised to n_states
    cols.assign((size_t)n_sites * n_leaves, 0);
    for (int r = 0; r < n_leaves; ++r) {
        const int32_t *row = codes + (size_t)r * n_sites;
        for (int64_t s = 0; s < n_sites; ++s) {
            const int32_t c = row[s];
            cols[(size_t)s * n_leaves + r] = (c >= 0 && c < n_states) ? (uint8_t)c : (uint8_t)n_states;
        }
    }
    site_to_pat.assign(n_sites, 0);
    first_site.clear();
    count.clear();
    if (compress) {
        std::unordered_map<uint64_t, std::vector<int64_t>> table;
        table.reserve((size_t)n_sites * 2);
        for (int64_t s = 0; s < n_sites; ++s) {
            const uint8_t *col = &cols[(size_t)s * n_leaves];
            uint64_t h = 0x2545f4914f6cdd1dull;
            int r = 0;
            for (; r + 8 <= n_leaves; r += 8) {
                uint64_t w;
                std::memcpy(&w, col + r, 8);
                h = mix64(h, w);
            }
            for (; r < n_leaves; ++r) h = mix64(h, col[r]);
            auto &bucket = table[h];
            int64_t found = -1;
            for (int64_t pid : bucket)
                if (std::memcmp(&cols[(size_t)first_site[pid] * n_leaves], col, n_leaves) == 0) {
                    found = pid;
                    break;
                }
            if (found < 0) {
                found = (int64_t)first_site.size();
                first_site.push_back(s);
                count.push_back(0.0);
                bucket.push_back(found);
            }
            count[found] += 1.0;
            site_to_pat[s] = found;
        }
        // Shard order.  Slices of the pattern list go to different GPUs, and what a GPU executes with clade tables grows with
        // the distinct sub-patterns its slice shows in every clade: in first-appearance order a 1/8 slice of the target
        // workload (M8, 500 taxa x 50 000 codons) executes 0.20 of the whole alignment's products, because every slice meets
        // nearly all of the common sub-patterns.  Patterns are therefore numbered by (consensus codon of the column, number
        // of leaves that differ from it, first appearance): columns that agree on most leaves -- and with them on most
        // clades' sub-patterns -- become neighbours, a 1/8 slice executes 0.145 (scripts/shard_orders.py; a key built from
        // the tree's own clades did no better).  A function of the alignment alone, so every rank numbers alike.
        const int64_t n_pat = (int64_t)first_site.size();
        std::vector<int64_t> key(n_pat), order(n_pat);
        std::vector<int> freq(n_states + 1);
        for (int64_t p = 0; p < n_pat; ++p) {
            const uint8_t *col = &cols[(size_t)first_site[p] * n_leaves];
            std::fill(freq.begin(), freq.end(), 0);
            for (int r = 0; r < n_leaves; ++r) ++freq[col[r]];
            int cons = n_states;
            for (int c = 0; c < n_states; ++c)
                if (freq[c] > (cons == n_states ? 0 : freq[cons])) cons = c;
            const int64_t n_diff = n_leaves - freq[cons];
            key[p] = ((int64_t)cons << 32) | n_diff;
            order[p] = p;
        }
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
        std::vector<int64_t> new_id(n_pat), first2(n_pat);
        std::vector<double> count2(n_pat);
        for (int64_t k = 0; k < n_pat; ++k) {
            new_id[order[k]] = k;
            first2[k] = first_site[order[k]];
            count2[k] = count[order[k]];
        }
        first_site.swap(first2);
        count.swap(count2);
        for (int64_t s = 0; s < n_sites; ++s) site_to_pat[s] = new_id[site_to_pat[s]];
    } else {
        first_site.resize(n_sites);
        count.assign(n_sites, 1.0);
        for (int64_t s = 0; s < n_sites; ++s) first_site[s] = site_to_pat[s] = s;
    }
}

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

void compress_alignment(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress, CompressedAlignment &ca)
{
    if (!codes || n_leaves < 1 || n_sites < 0) fail(CRAB_ERR_ARG, "crab_set_alignment: bad arguments");
    if (n_states < 1 || n_states > NS - 1) fail(CRAB_ERR_ARG, "n_states must be in [1, 63], got " + std::to_string(n_states));
    ca.n_leaves = n_leaves;
    ca.n_states = n_states;
    ca.n_sites = n_sites;
    compress_patterns(codes, n_leaves, n_sites, n_states, compress, ca.cols, ca.site_to_pat, ca.first_site, ca.count);
}

// Uploads patterns [pat_begin, pat_end) of a compressed alignment to the engine's device.
void upload_alignment(crab_engine *e, const CompressedAlignment &ca, int64_t pat_begin, int64_t pat_end)
{
    const int n_leaves = ca.n_leaves, n_states = ca.n_states;
    e->have_aln = false;
    e->n_states = n_states;
    e->aln_rows = n_leaves;
    e->n_sites = ca.n_sites;
    e->site_to_pat = ca.site_to_pat;
    e->n_pat_total = (int64_t)ca.first_site.size();
    if (pat_end < 0 || pat_end > e->n_pat_total) pat_end = e->n_pat_total;
    if (pat_begin < 0) pat_begin = 0;
    if (pat_begin > pat_end) pat_begin = pat_end;
    e->pat_begin = pat_begin;
    e->pat_end = pat_end;
    e->n_pat = pat_end - pat_begin;
    const int tile = prune_tile_width();
    e->n_pat_pad = ((e->n_pat + tile - 1) / tile) * tile;
    if (e->n_pat_pad == 0) e->n_pat_pad = tile;

    std::vector<uint8_t> img((size_t)n_leaves * e->n_pat_pad, (uint8_t)n_states);
    e->weights.assign(e->n_pat_pad, 0.0);
    for (int64_t p = 0; p < e->n_pat; ++p) {
        const uint8_t *col = &ca.cols[(size_t)ca.first_site[pat_begin + p] * n_leaves];
        for (int r = 0; r < n_leaves; ++r) img[(size_t)r * e->n_pat_pad + p] = col[r];
        e->weights[p] = ca.count[pat_begin + p];
    }
    HIP_CHECK(hipSetDevice(e->device));
    e->d_codes.ensure(img.size());
    e->d_weights.ensure(e->n_pat_pad);
    HIP_CHECK(hipMemcpyAsync(e->d_codes.p, img.data(), img.size(), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipMemcpyAsync(e->d_weights.p, e->weights.data(), e->n_pat_pad * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->h_codes.swap(img);
    e->clades = CladeTables{};
    e->dev_ops.clear();
    e->have_aln = true;
}

void load_alignment(crab_engine *e, const int32_t *codes, int n_leaves, int64_t n_sites, int n_states, int compress,
                    int64_t pat_begin, int64_t pat_end)
{
    e->have_aln = false;
    CompressedAlignment ca;
    compress_alignment(codes, n_leaves, n_sites, n_states, compress, ca);
    upload_alignment(e, ca, pat_begin, pat_end);
}

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

    // ---- partial updates (second half).  In a batch, a vector whose transition matrices differ from
    // vector 0's on exactly one branch -- a finite-difference row that moves one branch length, or one branch's omega in
    // the free-ratio model -- needs only the conditional likelihoods on the path from that branch to the root.  Vector
    // 0's walk saves the tiles such paths start from and pass by; those vectors then run path programs in a second
    // launch.  Same arithmetic in the same order as a whole walk: bit-identical results (tested). ----
    const int G = pl.n_groups;
    const int64_t n_tiles = e->n_pat_pad / prune_tile_width();
    const int Kg = pl.K / G, nn = e->n_nodes;
    std::vector<int32_t> vec_node(G, -1);   // per vector: the one node whose branch differs from vector 0's, or -1: whole walk
    int n_path = 0;
    size_t save_bytes = 0;
    if (G > 1 && prune_supports_paths() && !path_updates_disabled() && e->n_save_slots > 0) {
        for (int b = 1; b < G; ++b) {
            int node = -1, n_diff = 0;
            for (int v = 0; v < nn && n_diff < 2; ++v) {
                if (v == e->root || !e->reachable[v]) continue;
                bool diff = false;
                for (int k = 0; k < Kg && !diff; ++k) diff = (*pidx)[((size_t)b * Kg + k) * nn + v] != (*pidx)[(size_t)k * nn + v];
                if (diff) ++n_diff, node = v;
            }
            if (n_diff == 1) vec_node[b] = node, ++n_path;
        }
        save_bytes = (size_t)e->n_save_slots * Kg * (size_t)n_tiles * 4 * SAVE_TILE_BYTES;
        if (n_path < 4 || save_bytes > path_save_limit_bytes()) n_path = 0;
        if (n_path == 0) std::fill(vec_node.begin(), vec_node.end(), -1);
    }
    // ---- tree parts (a finer work quantum): a single evaluation with few work items per resident workgroup is cut into
    // subtree programs plus a root program (build_tree_parts) ----
    // ---- clade tables (subtree site repeats; build_clade_tables): every vector that walks the whole tree, except the base
    // vector of a batch with partial updates (the path programs want every internal edge's tiles per pattern) ----
    bool use_clades = false;
    std::vector<int32_t> tabled;          // classes (vector b, class k -> b * Kg + k) whose programs gather from clade tables
    if (prune_supports_paths() && e->clade_tables_on && !clade_tables_disabled() && e->n_pat > 0) {
        if (!e->clades.built) build_clade_tables(e);
        if (!e->clades.clades.empty()) {
            for (int b = 0; b < G; ++b)
                if (vec_node[b] < 0 && !(n_path > 0 && b == 0))
                    for (int k = 0; k < Kg; ++k) tabled.push_back(b * Kg + k);
            // (every tabled class owns its tables: CRAB_CLADE_TABLE_GB, default 32, bounds what a large batch may take)
            const double gb = (double)tabled.size() * e->clades.slots_per_class * (P_ELEMS * sizeof(double)) / 1073741824.0;
            const char *sv = std::getenv("CRAB_CLADE_TABLE_GB");
            if (gb > (sv && sv[0] ? std::atof(sv) : 32.0)) tabled.clear();
            use_clades = !tabled.empty();
        }
    }
    const int n_cl = use_clades ? (int)e->clades.clades.size() : 0;
    const int n_tab = (int)tabled.size();
    int part_target = 0;
    if (G == 1 && !use_clades && prune_supports_paths() && !tree_parts_disabled() && e->n_gemm_edges >= 8) {
        const int64_t slots = prune_grid_size(int64_t(1) << 40, e->device), n_work = n_tiles * pl.K;
        // measured (profiles/r2_tree_parts_timing.txt): a gain below ~3 items per resident workgroup (+15 % at config 3's
        // per-GPU share, +43 % at config 4's, +10 % at config 4 in full), a small loss at configs[1]'s 3.35
        if (5 * n_work < 16 * slots) {
            // aim at ~8 items per resident workgroup, at most 16 pieces, pieces of at least 4 products
            const int64_t pieces = std::min<int64_t>(16, std::max<int64_t>(2, (8 * slots + n_work - 1) / n_work));
            part_target = (int)std::max<int64_t>(4, (e->n_gemm_edges + pieces - 1) / pieces);
            if (e->tree_parts.target != part_target) build_tree_parts(e, part_target, e->tree_parts);
            if (e->tree_parts.cut.empty()) part_target = 0;
        }
    }
    const int n_cut = part_target > 0 ? (int)e->tree_parts.cut.size() : 0;
    if (n_cut > 0) save_bytes = (size_t)n_cut * Kg * (size_t)n_tiles * 4 * SAVE_TILE_BYTES;
    if (n_path > 0 || n_cut > 0) {
        try {
            e->d_save.ensure(save_bytes);
        } catch (const HipError &) {   // no room for the save area: whole walks
            (void)hipGetLastError();
            n_path = 0;
            part_target = 0;
            std::fill(vec_node.begin(), vec_node.end(), -1);
        }
    }
    const int n_parts = part_target > 0 ? n_cut : 0;
    // programs 0 .. K-1 write the site likelihoods; part / clade (k, i) is program K + k * n_parts + i
    const int n_prog = use_clades ? pl.K + n_tab * n_cl : pl.K * (1 + n_parts);

    // ---- uploads ----
    const size_t prog_stride = (size_t)dev_ops_padded((int64_t)e->ops.size());
    if (e->dev_ops.empty() || *pidx != e->dev_pidx || vec_node != e->dev_vec_node || part_target != e->dev_part_target ||
        use_clades != e->dev_clades || tabled != e->dev_tabled) {
        // the programs depend on the tree and on these tables only
        DevOp filler{};
        filler.w0 = OP_CONST;
        std::vector<DevOp> progs((size_t)n_prog * prog_stride, filler);
        std::vector<ProgHdr> hdrs(n_prog);
        std::vector<int32_t> lists;                 // first launch, then second launch
        auto header = [&](int prog, const ProgInfo &in, int save_cls, int no_root) {
            ProgHdr &h = hdrs[prog];
            h.n_ops = in.n_ops, h.first_gemm_op = in.first_gemm_op, h.first_leaf_row = in.first_leaf_row;
            h.first_leaf_op = in.first_leaf_op, h.second_leaf_row = in.second_leaf_row, h.third_leaf_row = in.third_leaf_row;
            h.second_leaf_mat = in.second_leaf_op, h.fourth_leaf_row = in.fourth_leaf_row;   // (op index for now: resolved below)
            h.save_cls = save_cls;
            h.no_root = no_root;
        };
        ProgInfo whole;
        whole.n_ops = (int)e->ops.size();
        whole.first_gemm_op = e->first_gemm_op, whole.first_leaf_row = e->first_leaf_row, whole.first_leaf_op = e->first_leaf_op;
        whole.second_leaf_row = e->second_leaf_row, whole.third_leaf_row = e->third_leaf_row;
        whole.second_leaf_op = e->second_leaf_op, whole.fourth_leaf_row = e->fourth_leaf_row;
        whole.n_gemm = (int)e->n_gemm_edges, whole.n_leaf = (int)e->n_leaf_edges;
        int n_first = 0, first_off = 0;
        e->dev_gemm_per_pat = e->dev_leaf_per_pat = 0;
        e->dev_gemm_extra = 0;
        struct ItemRef {
            int32_t prog, tile0, n;
        };
        std::vector<ItemRef> items;                 // clade launches: (program, first tile, tiles) per work item
        std::vector<int32_t> dep_recs;              // merged clade launch: per program, the tables it waits for
        // CRAB_CLADE_LEVEL_LAUNCHES=1: one launch per nesting level, as in round 2.  The merged launch needs the dynamic hand-out
        // (an item is only ever held by a workgroup that is running) and a device that deals workgroups to XCDs by number.
        const bool merged_clade_launch = !clade_level_launches_forced() && dynamic_items_enabled() && prune_device_has_all_xcds(e->device);
        if (n_parts > 0) {
            const TreeParts &tp = e->tree_parts;
            for (int k = 0; k < pl.K; ++k) {
                const int32_t *pk = pidx->data() + (size_t)k * nn;
                resolve_program(tp.root.ops, pk, false, progs.data() + (size_t)k * prog_stride);
                header(k, tp.root.info, k, 0);
                for (int i = 0; i < n_parts; ++i) {
                    const int prog = pl.K + k * n_parts + i;
                    resolve_program(tp.parts[i].ops, pk, false, progs.data() + (size_t)prog * prog_stride);
                    header(prog, tp.parts[i].info, k, 1);
                }
            }
            // launch 1: the parts, longest first, dealt into eight groups of similar total length (the kernel cuts its
            // work-item order into one chunk per XCD); launch 2: the root programs
            std::vector<int> order(n_parts);
            for (int i = 0; i < n_parts; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tp.parts[x].info.n_gemm > tp.parts[y].info.n_gemm; });
            for (int k = 0; k < pl.K; ++k)        // class-major (the matrices of one class stay together), longest parts first
                for (int i : order) lists.push_back(pl.K + k * n_parts + i);
            n_first = (int)lists.size();
            for (int k = 0; k < pl.K; ++k) lists.push_back(k);
            e->dev_gemm_per_pat = (int64_t)pl.K * e->n_gemm_edges;
            e->dev_leaf_per_pat = (int64_t)pl.K * e->n_leaf_edges;
        } else {
            std::vector<char> is_tabled(pl.K, 0);
            e->dev_clade_launches.clear();
            if (use_clades) {
                const CladeTables &ct = e->clades;
                // a program sees the tables it gathers from as the matrices of the clade roots: one patched node -> matrix table per
                // consumer (a clade, or the main program) and tabled class
                std::vector<int32_t> pk_c(nn);
                auto table_mat = [&](int t, int j) { return (int32_t)(M + t * ct.slots_per_class + ct.clades[j].slot_off); };
                for (int t = 0; t < n_tab; ++t) {
                    const int c = tabled[t];
                    is_tabled[c] = 1;
                    const int32_t *pk = pidx->data() + (size_t)c * nn;
                    for (int consumer = -1; consumer < n_cl; ++consumer) {
                        std::copy(pk, pk + nn, pk_c.begin());
                        for (int j = 0; j < n_cl; ++j)
                            if (ct.clades[j].parent == consumer) pk_c[ct.clades[j].root] = table_mat(t, j);
                        if (consumer < 0) {
                            resolve_program(ct.main.ops, pk_c.data(), false, progs.data() + (size_t)c * prog_stride);
                            header(c, ct.main.info, c % Kg, 0);
                        } else {
                            const PathProgram &pp = ct.clades[consumer].prog;
                            const int prog = pl.K + t * n_cl + consumer;
                            DevOp *po = progs.data() + (size_t)prog * prog_stride;
                            resolve_program(pp.ops, pk_c.data(), false, po);
                            po[pp.ops.size() - 1].mat = table_mat(t, consumer);   // the table store
                            header(prog, pp.info, c % Kg, 1);
                            // bit 0: leaf rows come from codes_alt; bits 2..: the clade's level (its slice of codes_alt); bit 1 (set
                            // below for a merged launch): the program waits for the tables of the clades inside it
                            hdrs[prog].alt_codes = 1 | ((ct.clades[consumer].level - 1) << 2);
                        }
                    }
                    e->dev_gemm_per_pat += ct.main.info.n_gemm;
                    e->dev_leaf_per_pat += ct.main.info.n_leaf;
                }
                // Work items: runs of consecutive tiles of one (class, clade) -- a cyclic program pays its prologue (item -> header
                // -> ops, codes -> columns -> operand image: a chain of dependent loads) once per item, later tiles start from what
                // the previous tile's last product requested.
                struct Item {
                    int32_t prog, tile0, n, level;
                    double w;
                };
                auto env_w = [](const char *name, double dflt) {
                    const char *sv = std::getenv(name);
                    return sv && sv[0] ? std::atof(sv) : dflt;
                };
                // predicted time of an item in products: a prologue, and per tile its products and a sixth of a product for the
                // rest (fitted to per-workgroup busy times, scripts/gpu_prof_phases.sh with -DCRAB_CLOCK)
                const double w_item = env_w("CRAB_W_ITEM", 0.4), w_tile = env_w("CRAB_W_TILE", 0.16);
                const int64_t slots = prune_grid_size(int64_t(1) << 40, e->device);
                auto items_of = [&](int t, int j, int64_t run_len, std::vector<Item> &out) {
                    const ProgInfo &in = ct.clades[j].prog.info;
                    const int64_t tiles = ct.clades[j].tiles;
                    const int64_t runs = in.cyclic ? (tiles + run_len - 1) / run_len : tiles;
                    for (int64_t r = 0; r < runs; ++r) {   // runs of equal length (+-1)
                        const int64_t lo = tiles * r / runs, hi = tiles * (r + 1) / runs;
                        out.push_back({(int32_t)(pl.K + t * n_cl + j), (int32_t)lo, (int32_t)(hi - lo), ct.clades[j].level,
                                       w_item + (double)(hi - lo) * (in.n_gemm + w_tile)});
                    }
                };
                int64_t run_env = 0;
                if (const char *sv = std::getenv("CRAB_CLADE_ITEM_TILES")) run_env = std::max(1, std::min(64, std::atoi(sv)));
                for (int t = 0; t < n_tab; ++t)
                    for (int j = 0; j < n_cl; ++j) e->dev_gemm_extra += (int64_t)ct.clades[j].prog.info.n_gemm * ct.clades[j].tiles * prune_tile_width();
                // A clade with more inner clades than a dependency record holds (6) cannot wait for them inside a launch.
                bool merge = merged_clade_launch;
                for (int j = 0; j < n_cl && merge; ++j) {
                    int inner = 0;
                    for (int c2 = 0; c2 < n_cl; ++c2) inner += ct.clades[c2].parent == j;
                    merge = inner <= 6;
                }
                dep_recs.clear();
                if (merge) {
                    // ONE launch for the clade programs of all levels.  A unit = one class's copy of an outermost clade with every
                    // clade nested inside it: tables are written and read inside a unit only.  Units are dealt to the XCDs by
                    // predicted work (largest first, each to the XCD with the least so far), so a table and the counter that
                    // announces it stay in one XCD's L2 -- nothing in the launch depends on coherence between XCDs.  An XCD's
                    // items are ordered by level, then longest first; the kernel hands them out in that order and an item whose
                    // program gathers from inner clades' tables waits for their counters (PruneArgs::deps, ::done): every item it
                    // waits for sits earlier in the same queue, and items of level 1 wait for nothing, so by induction over the
                    // levels every wait ends.
                    int64_t all_tiles = 0;
                    for (int j = 0; j < n_cl; ++j) all_tiles += ct.clades[j].tiles;
                    const int64_t run_len = run_env > 0 ? run_env : std::max<int64_t>(1, std::min<int64_t>(8, all_tiles * n_tab / (6 * slots)));
                    std::vector<int> outer_of(n_cl);
                    for (int j = 0; j < n_cl; ++j) {
                        int r = j;
                        while (ct.clades[r].parent >= 0) r = ct.clades[r].parent;
                        outer_of[j] = r;
                    }
                    struct Unit {
                        std::vector<Item> its;
                        double w = 0.0;
                    };
                    std::vector<Unit> units;
                    std::vector<int> unit_of(n_cl, -1);
                    for (int t = 0; t < n_tab; ++t) {
                        std::fill(unit_of.begin(), unit_of.end(), -1);
                        for (int j = 0; j < n_cl; ++j) {
                            int &u = unit_of[outer_of[j]];
                            if (u < 0) {
                                u = (int)units.size();
                                units.emplace_back();
                            }
                            const size_t before = units[u].its.size();
                            items_of(t, j, run_len, units[u].its);
                            for (size_t i = before; i < units[u].its.size(); ++i) units[u].w += units[u].its[i].w;
                        }
                    }
                    std::vector<int> by_w(units.size());
                    for (size_t u = 0; u < units.size(); ++u) by_w[u] = (int)u;
                    std::stable_sort(by_w.begin(), by_w.end(), [&](int p, int q) { return units[p].w > units[q].w; });
                    std::vector<std::vector<Item>> of_xcd(8);
                    double load[8] = {0};
                    for (int u : by_w) {
                        const int x = (int)(std::min_element(load, load + 8) - load);
                        load[x] += units[u].w;
                        of_xcd[x].insert(of_xcd[x].end(), units[u].its.begin(), units[u].its.end());
                    }
                    e->dev_clade_launches.assign(1, crab_engine::CladeLaunch{});
                    crab_engine::CladeLaunch &cl = e->dev_clade_launches[0];
                    cl.list_off = (int)lists.size();
                    cl.items_off = (int64_t)items.size();
                    for (int t = 0; t < n_tab; ++t)
                        for (int j = 0; j < n_cl; ++j) lists.push_back(pl.K + t * n_cl + j);
                    cl.n_entries = (int)lists.size() - cl.list_off;
                    cl.bounds[0] = 0;
                    for (int x = 0; x < 8; ++x) {
                        std::stable_sort(of_xcd[x].begin(), of_xcd[x].end(),
                                         [](const Item &p, const Item &q) { return p.level != q.level ? p.level < q.level : p.w > q.w; });
                        for (const Item &it : of_xcd[x]) items.push_back({it.prog, it.tile0, it.n});
                        cl.bounds[x + 1] = cl.bounds[x] + (int64_t)of_xcd[x].size();
                    }
                    cl.n_items = cl.bounds[8];
                    cl.grid = prune_grid_size(cl.n_items, e->device);
                    cl.merged = true;
                    // one record per program: [n, program x 6, tiles needed x 6, 0, 0, 0]; behind them, per program, the tiles it
                    // announces in all (PruneArgs::expect)
                    dep_recs.assign((size_t)n_prog * 17, 0);
                    for (int t = 0; t < n_tab; ++t)
                        for (int j = 0; j < n_cl; ++j) {
                            int32_t *d = dep_recs.data() + (size_t)(pl.K + t * n_cl + j) * 16;
                            dep_recs[(size_t)n_prog * 16 + pl.K + t * n_cl + j] = (int32_t)ct.clades[j].tiles;
                            for (int c2 = 0; c2 < n_cl; ++c2)
                                if (ct.clades[c2].parent == j) {
                                    d[1 + d[0]] = pl.K + t * n_cl + c2;
                                    d[7 + d[0]] = (int32_t)ct.clades[c2].tiles;
                                    ++d[0];
                                }
                            if (d[0] > 0) hdrs[pl.K + t * n_cl + j].alt_codes |= 2;
                        }
                    // What the kernel's waits rely on, checked here rather than found out on the device: every program's items sit
                    // in one XCD's chunk, after all items of the programs it waits for, whose tiles add up to what it waits for.
                    {
                        std::vector<int64_t> first(n_prog, -1), last(n_prog, -1), tiles_of(n_prog, 0);
                        std::vector<int> xcd_of(n_prog, -1);
                        bool ok = true;
                        for (int x = 0; x < 8; ++x)
                            for (int64_t i = cl.bounds[x]; i < cl.bounds[x + 1]; ++i) {
                                const ItemRef &it = items[(size_t)(cl.items_off + i)];
                                ok = ok && it.prog >= pl.K && it.prog < n_prog && (xcd_of[it.prog] < 0 || xcd_of[it.prog] == x);
                                if (!ok) break;
                                xcd_of[it.prog] = x;
                                if (first[it.prog] < 0) first[it.prog] = i;
                                last[it.prog] = i;
                                tiles_of[it.prog] += it.n;
                            }
                        for (int pr = pl.K; pr < n_prog && ok; ++pr) {
                            const int32_t *d = dep_recs.data() + (size_t)pr * 16;
                            ok = first[pr] >= 0 && d[0] >= 0 && d[0] <= 6 && tiles_of[pr] == ct.clades[(pr - pl.K) % n_cl].tiles;
                            for (int k = 0; k < d[0] && ok; ++k) {
                                const int q = d[1 + k];
                                ok = q >= pl.K && q < n_prog && xcd_of[q] == xcd_of[pr] && last[q] < first[pr] && tiles_of[q] == d[7 + k];
                            }
                        }
                        if (!ok) fail(CRAB_ERR_STATE, "clade work items: a table would be read before it is written");
                    }
                } else {
                    // one launch per level: class-major (one class's matrices stay together), clades with most products first
                    e->dev_clade_launches.assign(ct.n_levels, crab_engine::CladeLaunch{});
                    for (int l = 0; l < ct.n_levels; ++l) {
                        crab_engine::CladeLaunch &cl = e->dev_clade_launches[l];
                        cl.list_off = (int)lists.size();
                        cl.items_off = (int64_t)items.size();
                        std::vector<int> order;
                        for (int j = ct.level_begin[l]; j < ct.level_begin[l + 1]; ++j) order.push_back(j);
                        std::stable_sort(order.begin(), order.end(),
                                         [&](int x, int y) { return ct.clades[x].prog.info.n_gemm > ct.clades[y].prog.info.n_gemm; });
                        // runs as long as the launch can afford while it still has ~3 items per resident workgroup
                        int64_t level_tiles = 0;
                        for (int j : order) level_tiles += ct.clades[j].tiles;
                        const int64_t run_len = run_env > 0 ? run_env : std::max<int64_t>(1, std::min<int64_t>(8, level_tiles * n_tab / (3 * slots)));
                        std::vector<Item> its;
                        for (int t = 0; t < n_tab; ++t)
                            for (int j : order) {
                                lists.push_back(pl.K + t * n_cl + j);
                                items_of(t, j, run_len, its);
                            }
                        cl.n_entries = (int)lists.size() - cl.list_off;
                        cl.n_items = (int64_t)its.size();
                        // One chunk of the launch's items per XCD, of equal predicted work (class-major order: the workgroups that
                        // share an L2 walk the same matrices); inside a chunk the longest items first -- the kernel hands a chunk's
                        // items out in order to whichever of the XCD's workgroups asks next: the tail of the launch is a short item.
                        double total = 0.0, acc = 0.0;
                        for (const Item &it : its) total += it.w;
                        int x = 1;
                        cl.bounds[0] = 0;
                        for (size_t i = 0; i < its.size() && x < 8; ++i) {
                            acc += its[i].w;
                            while (x < 8 && acc >= total * x / 8.0) cl.bounds[x++] = (int64_t)i + 1;
                        }
                        while (x <= 8) cl.bounds[x++] = (int64_t)its.size();
                        cl.grid = prune_grid_size((int64_t)its.size(), e->device);
                        for (int c8 = 0; c8 < 8; ++c8)
                            std::stable_sort(its.begin() + cl.bounds[c8], its.begin() + cl.bounds[c8 + 1],
                                             [](const Item &p, const Item &q) { return p.w > q.w; });
                        for (const Item &it : its) items.push_back({it.prog, it.tile0, it.n});
                    }
                }
            }
            first_off = (int)lists.size();
            for (int b = 0; b < G; ++b) {
                if (is_tabled[(size_t)b * Kg]) continue;
                const PathProgram *pp = nullptr;
                if (vec_node[b] >= 0) {
                    auto it = e->path_programs.find(vec_node[b]);
                    if (it == e->path_programs.end()) {
                        it = e->path_programs.emplace(vec_node[b], PathProgram{}).first;
                        build_path_program(e, vec_node[b], it->second);
                    }
                    pp = &it->second;
                }
                for (int k = 0; k < Kg; ++k) {
                    const int c = b * Kg + k;
                    resolve_program(pp ? pp->ops : e->ops, pidx->data() + (size_t)c * nn, n_path > 0 && b == 0, progs.data() + (size_t)c * prog_stride);
                    header(c, pp ? pp->info : whole, k, 0);
                }
                e->dev_gemm_per_pat += (int64_t)Kg * (pp ? pp->info.n_gemm : whole.n_gemm);
                e->dev_leaf_per_pat += (int64_t)Kg * (pp ? pp->info.n_leaf : whole.n_leaf);
            }
            // launch 1: whole walks; launch 2: path programs, dealt into eight groups of similar total length
            for (int b = 0; b < G; ++b)
                if (vec_node[b] < 0)
                    for (int k = 0; k < Kg; ++k) lists.push_back(b * Kg + k);
            n_first = (int)lists.size() - first_off;
            std::vector<int> order;
            for (int b = 0; b < G; ++b)
                if (vec_node[b] >= 0) order.push_back(b);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return hdrs[(size_t)x * Kg].n_ops > hdrs[(size_t)y * Kg].n_ops; });
            for (int k = 0; k < Kg; ++k)          // class-major, longest paths first
                for (int b : order) lists.push_back(b * Kg + k);
        }
        // One chunk of the work-item order per XCD, of equal work rather than equal item count: the programs of a launch
        // differ in length (a product costs about ten leaf edges; every item pays a fixed prologue).
        auto bounds = [&](int lo, int hi, int64_t *out) {
            std::vector<double> w(hi - lo);
            double total = 0.0;
            for (int i = lo; i < hi; ++i) {
                const ProgHdr &h = hdrs[lists[i]];
                int n_gemm = 0;
                const DevOp *po = progs.data() + (size_t)lists[i] * prog_stride;
                for (int j = 0; j < h.n_ops; ++j) n_gemm += (po[j].w0 & 3) == OP_GEMM;
                w[i - lo] = 1.0 + n_gemm + 0.1 * (h.n_ops - n_gemm);
                total += w[i - lo] * (double)n_tiles;
            }
            out[0] = 0;
            double acc = 0.0;
            int x = 1;
            for (int i = lo; i < hi && x < 8; ++i) {
                // cut inside this program's run of tiles wherever a boundary falls
                while (x < 8 && acc + w[i - lo] * (double)n_tiles >= total * x / 8.0) {
                    const double need = total * x / 8.0 - acc;
                    out[x++] = (int64_t)(i - lo) * n_tiles + std::min<int64_t>(n_tiles, (int64_t)std::llround(need / w[i - lo]));
                }
                acc += w[i - lo] * (double)n_tiles;
            }
            while (x < 8) out[x++] = (int64_t)(hi - lo) * n_tiles;
            out[8] = (int64_t)(hi - lo) * n_tiles;
            for (int y = 1; y <= 8; ++y) out[y] = std::max(out[y], out[y - 1]);
        };
        bounds(first_off, first_off + n_first, e->dev_bounds_first);
        bounds(first_off + n_first, (int)lists.size(), e->dev_bounds_second);
        for (int pr = 0; pr < n_prog; ++pr) {   // what an item's prologue needs from the ops, so that it need not wait for them
            ProgHdr &h = hdrs[pr];
            const DevOp *po = progs.data() + (size_t)pr * prog_stride;
            h.first_gemm_mat = h.first_gemm_op >= 0 ? po[h.first_gemm_op].mat : -1;
            h.first_leaf_mat = h.first_leaf_op >= 0 ? po[h.first_leaf_op].mat : -1;
            h.second_leaf_mat = h.second_leaf_mat >= 0 ? po[h.second_leaf_mat].mat : -1;
        }
        e->dev_clades = use_clades;
        e->dev_tabled = tabled;
        e->dev_first_off = first_off;
        // an item record = its program's header + (program, first tile, tiles): one scalar load per item on the device
        std::vector<ProgHdr> item_recs(items.size());
        for (size_t i = 0; i < items.size(); ++i) {
            item_recs[i] = hdrs[items[i].prog];
            item_recs[i].prog = items[i].prog, item_recs[i].tile0 = items[i].tile0, item_recs[i].item_tiles = items[i].n;
        }
        e->dev_n_prog = n_prog;
        e->d_deps.ensure(dep_recs.size() + 16);
        if (!dep_recs.empty())
            HIP_CHECK(hipMemcpyAsync(e->d_deps.p, dep_recs.data(), dep_recs.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        e->d_items.ensure(item_recs.size() + 1);
        if (!item_recs.empty())
            HIP_CHECK(hipMemcpyAsync(e->d_items.p, item_recs.data(), item_recs.size() * sizeof(ProgHdr), hipMemcpyHostToDevice, e->stream));
        e->dev_pidx = *pidx;
        e->dev_vec_node = vec_node;
        e->dev_part_target = part_target;
        e->dev_n_first = n_first;
        e->dev_n_second = (int)lists.size() - first_off - n_first;
        e->d_ops.ensure(progs.size());
        e->d_hdr.ensure(hdrs.size());
        e->d_cls.ensure(lists.size() + 1);
        HIP_CHECK(hipMemcpyAsync(e->d_ops.p, progs.data(), progs.size() * sizeof(DevOp), hipMemcpyHostToDevice, e->stream));
        HIP_CHECK(hipMemcpyAsync(e->d_hdr.p, hdrs.data(), hdrs.size() * sizeof(ProgHdr), hipMemcpyHostToDevice, e->stream));
        HIP_CHECK(hipMemcpyAsync(e->d_cls.p, lists.data(), lists.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        HIP_CHECK(hipStreamSynchronize(e->stream));  // the sources are pageable
        e->dev_ops.swap(progs);
    }
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
    launch_build_p(pool.d, d_eig_id, d_teff, e->d_P.p, M, n, true, e->stream);

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
    // Every launch of the pruning kernel hands its items out through eight counters of its own (PruneArgs::work_ctr;
    // CRAB_STATIC_ITEMS=1: items dealt round-robin instead, as in rounds 1-2).
    constexpr int MAX_PRUNE_LAUNCHES = 16;
    // Behind them in the same buffer, cleared with them: the fault flag and the per-program table counters of a merged clade launch.
    constexpr int SYNC_AT = 8 * MAX_PRUNE_LAUNCHES;   // (PruneArgs::sync: the flag, 15 unused, the counters)
    const bool dynamic_items = dynamic_items_enabled();
    const bool merged = use_clades && !e->dev_clade_launches.empty() && e->dev_clade_launches[0].merged;
    if (merged && !dynamic_items) fail(CRAB_ERR_ARG, "merged clade launch without dynamic work items");
    if (dynamic_items) {
        const size_t n_ctr = (size_t)SYNC_AT + 16 + (merged ? (size_t)e->dev_n_prog : 0);
        e->d_work_ctr.ensure(n_ctr);
        if ((int)e->dev_clade_launches.size() + 2 > MAX_PRUNE_LAUNCHES) fail(CRAB_ERR_ARG, "too many clade levels");
        HIP_CHECK(hipMemsetAsync(e->d_work_ctr.p, 0, n_ctr * sizeof(int32_t), e->stream));
    }
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
            pa.expect = e->d_deps.p + (size_t)e->dev_n_prog * 16;
            pa.n_expect = e->dev_n_prog;
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

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// the codon graph: ordered single-nucleotide neighbour pairs among the 61 sense codons in PAML order
// (codon.rs:63-87,167-220), row by row with ascending column
// ---------------------------------------------------------------------------------------------
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

}  // namespace

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
                const int launch = cl.parent < 0 ? ct.n_levels : ct.clades[cl.parent].level - 1;
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

// =============================================================================================
// one-shot layer: the seven PyO3 signatures on top of a handle cache
// =============================================================================================
namespace {

struct CacheKey {  // contents, not addresses: callers may re-pack the same alignment on every call
    int device;
    int n_leaves;
    int64_t n_sites;
    int n_states;
    uint64_t data_hash[2];                       // 128 bits over the codes (the tree is compared entry by entry)
    std::vector<int32_t> parents, node_ids, leaf_ids;
    bool operator==(const CacheKey &o) const
    {
        return device == o.device && n_leaves == o.n_leaves && n_sites == o.n_sites && n_states == o.n_states &&
               data_hash[0] == o.data_hash[0] && data_hash[1] == o.data_hash[1] && parents == o.parents &&
               node_ids == o.node_ids && leaf_ids == o.leaf_ids;
    }
};

struct CacheEntry {
    CacheKey key;
    crab_engine *engine = nullptr;
    std::mutex busy;                             // one evaluation at a time per engine; other entries run alongside
    ~CacheEntry()
    {
        if (engine) crab_destroy(engine);        // runs when the last user lets go: eviction never pulls an engine from under a call
    }
};

std::mutex g_cache_mutex;                        // guards the list only, never held across an evaluation
// most recent first; never destroyed: at process exit the HIP runtime may already be gone, so engines still cached then are
// left to the operating system (crab_clear_cache releases them in an orderly way)
std::list<std::shared_ptr<CacheEntry>> &g_cache = *new std::list<std::shared_ptr<CacheEntry>>;
constexpr size_t CACHE_MAX = 8;

}  // namespace
static void hash_lanes(const void *p, size_t bytes, uint64_t seed, uint64_t h[4], size_t *tail_from)
{
    const unsigned char *b = (const unsigned char *)p;
    h[0] = seed, h[1] = seed ^ 0x9e3779b97f4a7c15ull, h[2] = seed + 0x632be59bd9b4e019ull, h[3] = ~seed;
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        std::memcpy(w, b + i, 32);
        h[0] = (h[0] ^ w[0]) * 0x9fb21c651e98df25ull; h[0] ^= h[0] >> 29;
        h[1] = (h[1] ^ w[1]) * 0xd6e8feb86659fd93ull; h[1] ^= h[1] >> 31;
        h[2] = (h[2] ^ w[2]) * 0xc2b2ae3d27d4eb4full; h[2] ^= h[2] >> 27;
        h[3] = (h[3] ^ w[3]) * 0x165667b19e3779f9ull; h[3] ^= h[3] >> 33;
    }
    *tail_from = i;
}

uint64_t hash_words(const void *p, size_t bytes, uint64_t seed)
{
    uint64_t l[4];
    size_t i;
    hash_lanes(p, bytes, seed, l, &i);
    uint64_t h = mix64(mix64(l[0], l[1]), mix64(l[2], l[3]));
    for (const unsigned char *b = (const unsigned char *)p; i < bytes; ++i) h = mix64(h, b[i]);
    return mix64(h, bytes);
}

// two 64-bit digests of the same pass (the second folds the four lanes in another order with other constants)
void hash_words2(const void *p, size_t bytes, uint64_t seed, uint64_t out[2])
{
    uint64_t l[4];
    size_t i;
    hash_lanes(p, bytes, seed, l, &i);
    uint64_t h = mix64(mix64(l[0], l[1]), mix64(l[2], l[3]));
    uint64_t g = mix64(mix64(l[3] + 0x2545f4914f6cdd1dull, l[1]), mix64(l[0] ^ 0xff51afd7ed558ccdull, l[2]));
    for (const unsigned char *b = (const unsigned char *)p; i < bytes; ++i) h = mix64(h, b[i]), g = mix64(g ^ 0x5bd1e995, b[i]);
    out[0] = mix64(h, bytes), out[1] = mix64(g, ~(uint64_t)bytes);
}
namespace {

// device = -1 in the one-shot functions: the device list of the environment variable CRAB_DEVICES ("0,1,2,3": one
// multi-device handle over those GPUs; unset or empty: device 0)
std::vector<int> devices_from_env()
{
    std::vector<int> out;
    if (const char *sv = std::getenv("CRAB_DEVICES")) {
        const char *p = sv;
        while (*p) {
            char *end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            out.push_back((int)v);
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
    }
    if (out.empty()) out.push_back(0);
    return out;
}

// Returns the cache entry for this (alignment, tree), creating engine and entry on first sight.  Hashing happens before
// the list lock is taken; a new engine is built under it (rare: once per alignment).
std::shared_ptr<CacheEntry> cached_engine(int device, int n_states, int n_nodes, const int32_t *node_ids, const int32_t *parents,
                                          const int32_t *codes, int n_leaves, int64_t n_sites, const int32_t *leaf_ids, int *status)
{
    *status = CRAB_OK;
    if (!codes || !parents || n_nodes < 1 || n_leaves < 1 || n_sites < 0) {
        g_last_error = "bad arguments";
        *status = CRAB_ERR_ARG;
        return nullptr;
    }
    std::vector<int> devs;
    if (device < 0) {
        devs = devices_from_env();
        device = -1 - (int)(hash_words(devs.data(), devs.size() * sizeof(int), 5) & 0x3fffffff);   // one key per device list
    }
    CacheKey key{device, n_leaves, n_sites, n_states, {0, 0}, {}, {}, {}};
    key.parents.assign(parents, parents + n_nodes);
    if (node_ids) key.node_ids.assign(node_ids, node_ids + n_nodes);
    if (leaf_ids) key.leaf_ids.assign(leaf_ids, leaf_ids + n_leaves);
    // the stateless boundary hands the alignment over on every call (4 MB at C2): hash it in 256 KiB chunks on the
    // host pool, then hash the chunk hashes
    {
        const size_t bytes = (size_t)n_leaves * n_sites * sizeof(int32_t), chunk = 256 * 1024;
        const int n_chunks = (int)((bytes + chunk - 1) / chunk);
        if (n_chunks <= 1) {
            hash_words2(codes, bytes, 1, key.data_hash);
        } else {
            std::vector<uint64_t> part(2 * (size_t)n_chunks);
            parallel_for(n_chunks, [&](int i) {
                const size_t lo = (size_t)i * chunk;
                hash_words2((const char *)codes + lo, std::min(chunk, bytes - lo), 1 + (uint64_t)i, &part[2 * (size_t)i]);
            });
            hash_words2(part.data(), part.size() * sizeof(uint64_t), bytes, key.data_hash);
        }
    }
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
        if ((*it)->key == key) {
            g_cache.splice(g_cache.begin(), g_cache, it);
            return g_cache.front();
        }
    crab_engine *e = nullptr;
    int rc = devs.size() > 1 ? crab_create_multi(devs.data(), (int)devs.size(), &e) : crab_create(devs.empty() ? device : devs[0], &e);
    if (rc == CRAB_OK) {
        rc = crab_set_tree(e, n_nodes, node_ids, parents, leaf_ids, n_leaves);
        if (rc == CRAB_OK) rc = crab_set_alignment(e, codes, n_leaves, n_sites, n_states, 1, 0, -1);
        if (rc != CRAB_OK) {
            g_last_error = e->err;
            crab_destroy(e);
            e = nullptr;
        }
    }
    *status = rc;
    if (!e) return nullptr;
    auto entry = std::make_shared<CacheEntry>();
    entry->key = std::move(key);
    entry->engine = e;
    g_cache.push_front(entry);
    while (g_cache.size() > CACHE_MAX) g_cache.pop_back();   // an evicted engine lives until its current call returns
    return entry;
}

template <typename F>
int with_cached(int device, int n_states, int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *codes,
                int n_leaves, int64_t n_sites, const int32_t *leaf_ids, F &&call)
{
    int status = CRAB_OK;
    std::shared_ptr<CacheEntry> entry = cached_engine(device, n_states, n_nodes, node_ids, parents, codes, n_leaves, n_sites, leaf_ids, &status);
    if (!entry) return status;
    std::lock_guard<std::mutex> lk(entry->busy);
    const int rc = call(entry->engine);
    if (rc != CRAB_OK) g_last_error = entry->engine->err;
    return rc;
}

}  // namespace

extern "C" {

int crab_compute_log_likelihood(int device, int n, const double *q, const double *pi, int n_nodes, const int32_t *node_ids,
                                const int32_t *parents, const double *branch_lengths, const int32_t *codes, int n_leaves,
                                int64_t n_sites, const int32_t *leaf_node_ids, double *lnL)
{
    const double one = 1.0;
    return with_cached(device, n, n_nodes, node_ids, parents, codes, n_leaves, n_sites, leaf_node_ids,
                       [&](crab_engine *e) { return crab_eval_site_classes(e, 1, q, &one, pi, branch_lengths, 0, lnL, nullptr); });
}

int crab_compute_site_class_log_likelihood(int device, int K, int n, const double *q_matrices, const double *proportions,
                                           const double *pi, int n_nodes, const int32_t *node_ids, const int32_t *parents,
                                           const double *branch_lengths, const int32_t *codes, int n_leaves, int64_t n_sites,
                                           const int32_t *leaf_node_ids, double *lnL)
{
    return with_cached(device, n, n_nodes, node_ids, parents, codes, n_leaves, n_sites, leaf_node_ids, [&](crab_engine *e) {
        return crab_eval_site_classes(e, K, q_matrices, proportions, pi, branch_lengths, 0, lnL, nullptr);
    });
}

int crab_compute_site_log_likelihoods_by_class(int device, int K, int n, const double *q_matrices, const double *pi,
                                               int n_nodes, const int32_t *node_ids, const int32_t *parents,
                                               const double *branch_lengths, const int32_t *codes, int n_leaves,
                                               int64_t n_sites, const int32_t *leaf_node_ids, double *out)
{
    return with_cached(device, n, n_nodes, node_ids, parents, codes, n_leaves, n_sites, leaf_node_ids, [&](crab_engine *e) {
        return crab_eval_site_lnl_by_class(e, K, q_matrices, pi, branch_lengths, 0, out);
    });
}

int crab_compute_branch_site_log_likelihood(int device, int n, const double *q_omega0, const double *q_omega1,
                                            const double *q_omega2, double qfactor_back, double qfactor_fore, double omega0,
                                            double omega2, const double *site_class_freqs, const double *pi, int n_nodes,
                                            const int32_t *node_ids, const int32_t *parents, const double *branch_lengths,
                                            const uint8_t *branch_labels, int n_labels, const int32_t *codes, int n_leaves,
                                            int64_t n_sites, const int32_t *leaf_node_ids, double *lnL)
{
    if (!q_omega0 || !q_omega1 || !q_omega2 || n < 1) {
        g_last_error = "bad arguments";
        return CRAB_ERR_ARG;
    }
    std::vector<double> Q3((size_t)3 * n * n);
    std::memcpy(Q3.data(), q_omega0, sizeof(double) * n * n);
    std::memcpy(Q3.data() + (size_t)n * n, q_omega1, sizeof(double) * n * n);
    std::memcpy(Q3.data() + (size_t)2 * n * n, q_omega2, sizeof(double) * n * n);
    return with_cached(device, n, n_nodes, node_ids, parents, codes, n_leaves, n_sites, leaf_node_ids, [&](crab_engine *e) {
        return crab_eval_branch_site(e, Q3.data(), qfactor_back, qfactor_fore, omega0, omega2, site_class_freqs, pi,
                                     branch_lengths, branch_labels, n_labels, 0, lnL, nullptr);
    });
}

int crab_compute_log_likelihood_branch(int device, int n, const int32_t *codes, int n_leaves, int64_t n_sites,
                                       const int32_t *parent_indices, int n_nodes, const int32_t *leaf_node_ids,
                                       const double *Q_matrices, int n_q, const double *branch_lengths, int n_branch_lengths,
                                       const double *pi, double *lnL)
{
    return with_cached(device, n, n_nodes, nullptr, parent_indices, codes, n_leaves, n_sites, leaf_node_ids, [&](crab_engine *e) {
        return crab_eval_branch(e, n_q, Q_matrices, pi, branch_lengths, n_branch_lengths, 0, lnL, nullptr);
    });
}

void crab_clear_cache(void)
{
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_cache.clear();                             // engines still inside a call are destroyed when it returns
}

}  // extern "C"
