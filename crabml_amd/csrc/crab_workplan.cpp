// Host side of libcrabml_amd.so: one evaluation cut into launches and work items (see crab_host.h).
#include "crab_host.h"

namespace crab_host {

namespace {
struct ItemRef {   // a work item of a clade launch: (program, first tile, tiles)
    int32_t prog, tile0, n;
};
}  // namespace

// The clade programs of one evaluation as launches of work items: ONE launch whose items wait for the tables they gather from
// (merged_allowed, and no clade with more inner clades than a dependency record holds), or one launch per nesting level.
// Appends to the program list and the item list, fills the dependency records and the waiting bit of the headers, and leaves
// the launches in e->dev_clade_launches.  n_cl clades x n_tab tabled classes; clade (t, j) is program K + t * n_cl + j of n_prog.
static void schedule_clade_launches(crab_engine *e, const CladeTables &ct, int K, int n_tab, int n_prog, bool merged_allowed,
                                    std::vector<int32_t> &lists, std::vector<ItemRef> &items, std::vector<int32_t> &dep_recs,
                                    std::vector<ProgHdr> &hdrs)
{
    const int n_cl = (int)ct.clades.size();
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
            out.push_back({(int32_t)(K + t * n_cl + j), (int32_t)lo, (int32_t)(hi - lo), ct.clades[j].level,
                           w_item + (double)(hi - lo) * (in.n_gemm + w_tile)});
        }
    };
    int64_t run_env = 0;
    if (const char *sv = std::getenv("CRAB_CLADE_ITEM_TILES")) run_env = std::max(1, std::min(64, std::atoi(sv)));
    for (int t = 0; t < n_tab; ++t)
        for (int j = 0; j < n_cl; ++j) e->dev_gemm_extra += (int64_t)ct.clades[j].prog.info.n_gemm * ct.clades[j].tiles * prune_tile_width();
    // A clade with more inner clades than a dependency record holds (6) cannot wait for them inside a launch.
    bool merge = merged_allowed;
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
            for (int j = 0; j < n_cl; ++j) lists.push_back(K + t * n_cl + j);
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
        // one record per program: [n, program x 6, tiles needed x 6, 0, 0, 0]
        dep_recs.assign((size_t)n_prog * 16, 0);
        e->dev_clade_tiles = (int32_t)(all_tiles * n_tab);   // what the launch announces in all (PruneArgs::expect_tiles)
        for (int t = 0; t < n_tab; ++t)
            for (int j = 0; j < n_cl; ++j) {
                int32_t *d = dep_recs.data() + (size_t)(K + t * n_cl + j) * 16;
                for (int c2 = 0; c2 < n_cl; ++c2)
                    if (ct.clades[c2].parent == j) {
                        d[1 + d[0]] = K + t * n_cl + c2;
                        d[7 + d[0]] = (int32_t)ct.clades[c2].tiles;
                        ++d[0];
                    }
                if (d[0] > 0) hdrs[K + t * n_cl + j].alt_codes |= 2;
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
                    ok = ok && it.prog >= K && it.prog < n_prog && (xcd_of[it.prog] < 0 || xcd_of[it.prog] == x);
                    if (!ok) break;
                    xcd_of[it.prog] = x;
                    if (first[it.prog] < 0) first[it.prog] = i;
                    last[it.prog] = i;
                    tiles_of[it.prog] += it.n;
                }
            for (int pr = K; pr < n_prog && ok; ++pr) {
                const int32_t *d = dep_recs.data() + (size_t)pr * 16;
                ok = first[pr] >= 0 && d[0] >= 0 && d[0] <= 6 && tiles_of[pr] == ct.clades[(pr - K) % n_cl].tiles;
                for (int k = 0; k < d[0] && ok; ++k) {
                    const int q = d[1 + k];
                    ok = q >= K && q < n_prog && xcd_of[q] == xcd_of[pr] && last[q] < first[pr] && tiles_of[q] == d[7 + k];
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
                    lists.push_back(K + t * n_cl + j);
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

// Decides how one evaluation's work is cut -- partial updates of a batch, clade tables, tree parts -- and, when the programs
// that follow from it differ from those on the device, resolves, schedules and uploads them (work items, per-XCD chunks,
// dependency records of a merged clade launch).  pidx: the (class, node) -> matrix table; M: matrices of the evaluation.
WorkSplit plan_device_work(crab_engine *e, const EvalPlan &pl, const std::vector<int32_t> *pidx, int M)
{
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
        std::vector<ItemRef> items;                 // clade launches: (program, first tile, tiles) per work item
        std::vector<int32_t> dep_recs;              // merged clade launch: per program, the tables it waits for
        // CRAB_CLADE_LEVEL_LAUNCHES=1: one launch per nesting level, as in round 2.  The merged launch needs the dynamic hand-out
        // (an item is only ever held by a workgroup that is running) and a device that deals workgroups to XCDs by number.
        const bool merged_clade_launch = !clade_level_launches_forced() && !e->merged_clade_launch_off && dynamic_items_enabled() && prune_device_has_all_xcds(e->device);
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
                schedule_clade_launches(e, ct, pl.K, n_tab, n_prog, merged_clade_launch, lists, items, dep_recs, hdrs);
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
    WorkSplit ws;
    ws.G = G, ws.Kg = Kg, ws.n_path = n_path, ws.n_parts = n_parts, ws.n_cl = n_cl, ws.n_tab = n_tab;
    ws.n_tiles = n_tiles;
    ws.use_clades = use_clades;
    return ws;
}

}  // namespace crab_host
