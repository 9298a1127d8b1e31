// Host side of libcrabml_amd.so: the stateless entry points over a cache of handles (see crab_host.h).
#include "crab_host.h"

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

