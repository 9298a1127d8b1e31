// Host side of libcrabml_amd.so: tree -> device programs (see crab_host.h).
#include "crab_host.h"

namespace crab_host {

// ---------------------------------------------------------------------------------------------
// tree -> program
// ---------------------------------------------------------------------------------------------
// Links the products of a program (a GEMM op's arg is the index of the next GEMM op, whose operand image it stages)
// and lays down the leaf-edge prefetch schedule.  Used for the whole-tree program and for path programs alike.
// `cyclic` (clade programs): the program is laid out to run over consecutive pattern tiles in one work item -- beyond its last
// leaf edge the code queue continues with the first leaf edges of the next tile (ROW_NEXT_TILE), and the last product requests
// their columns and stages the first product's operand image while it runs (F_NEXT_TILE).
void finalise_program(std::vector<TreeOp> &ops, ProgInfo &info, bool cyclic)
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


}  // namespace crab_host
