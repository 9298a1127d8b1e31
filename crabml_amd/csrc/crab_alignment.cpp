// Host side of libcrabml_amd.so: site patterns and content digests (see crab_host.h).
#include "crab_host.h"

namespace crab_host {

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


}  // namespace crab_host

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
