// Device-side interface of the likelihood engine: data layouts and kernel launchers.
// All launchers enqueue on `stream` and never synchronise or allocate.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "crab_eigen.h"

namespace crab {

// ---- layouts ------------------------------------------------------------------------------
// A transition matrix P (64 x 64 padded, P[i][j] = Pr(j at child | i at parent)) is kept twice, 64 KiB in all:
//
//  operand image (doubles [0, 4096)): the A-operand order of v_mfma_f64_4x4x4_4b_f64.  P is cut into 4x4
//     tiles (I = i/4, s = j/4); a lane l of the wave supplies A[i' = l&3][k = l>>4] of its tile (the same
//     value in all four 4-lane blocks), so a tile is 16 doubles ordered (k*4 + i').  Tiles s and s+1 are
//     interleaved so that one 16-byte LDS read feeds two MFMAs:
//         element (i, j) lives at  (((i>>2)*8 + (j>>3))*16 + (j&3)*4 + (i&3))*2 + ((j>>2)&1)
//  gather image (doubles [4096, 8192)): column j of P (512 contiguous bytes) cut into eight 64-byte pieces; piece q
//     holds, for each of the four lanes g = i&3 that share a pattern, the two states of register pair q:
//         element (i, j) lives at  4096 + j*64 + (i>>3)*8 + (i&3)*2 + ((i>>2)&1)
//     so a leaf edge is eight 16-byte loads per lane and each load instruction touches one half cache line per
//     pattern (16 lines per wave instruction, not 64).
//
// Column n_states (61) holds the row sums sum_j P[i][j] (used for missing data); rows and columns
// >= n_states are otherwise zero.
constexpr int P_IMAGE = NS * NS;       // doubles per image (32 KiB)
constexpr int P_ELEMS = 2 * P_IMAGE;   // doubles per matrix, both images

__host__ __device__ inline int p_operand_index(int i, int j)
{
    return ((((i >> 2) * 8 + (j >> 3)) * 16) + (j & 3) * 4 + (i & 3)) * 2 + ((j >> 2) & 1);
}
__host__ __device__ inline int p_gather_index(int i, int j) { return P_IMAGE + j * 64 + (i >> 3) * 8 + (i & 3) * 2 + ((i >> 2) & 1); }

// Conditional likelihoods never touch HBM.  Inside the pruning kernel a wave holds a
// [64 states x 16 patterns] tile per pattern group: lane l owns pattern (l&15) and the 16 states
// 4*idx + (l>>4), idx = 0..15.  That is the C/D layout of the 4x4x4 MFMA (D[blk][i][j] in lane
// 16*i + 4*blk + j) and also its B-operand layout (B[blk][k][j] in lane 16*k + 4*blk + j), so a parent's
// product consumes the child's tile straight from registers, register idx being k-step idx.

// ---- tree program ---------------------------------------------------------------------------
enum : int32_t {
    OP_GEMM = 0, OP_LEAF = 1, OP_CONST = 2,
    OP_LOAD = 3   // the current tile is (or, with F_MUL_PROD, is multiplied by) save slot `sv`, written by an earlier launch
                  // (path programs, root parts); with F_SAVE the other way round: the current tile goes to slot `sv` (tree parts)
};
enum : int32_t {
    F_MUL_PROD = 1,   // multiply the new term into the running product of this node
    F_POP_MUL = 2,    // multiply the new term with stack[slot] (the product saved before a sibling subtree)
    F_PUSH = 4,       // after this op, save the running product in stack[slot]
    F_FIN = 8,        // after this op the node is complete: it becomes the operand of the next GEMM
    F_RESCALE = 16,   // after this op, normalise the running product (every second tree level; long runs of children)
    // leaf-edge prefetch schedule (used by kernels that keep a prefetch tile): the columns of the next leaf edge are
    // requested while a product runs, or at the end of the op just before that leaf edge
    F_PRE_EARLY = 32, // GEMM: before the product starts, request the columns of leaf edge pre_node
    F_PRE_LATE = 64,  // any op: at its end, request the columns of leaf edge pre_node (the next op is that leaf edge)
    F_PRE_DYING = 128,  // GEMM: also request the columns of leaf edge pre_node_b into the registers of the product's B
                        // operand as they die, k-tile by k-tile; they are then in the landing tile of the ops that follow
    F_IN_LANDING = 256, // LEAF (with F_MUL_PROD): the columns were requested by the preceding product's F_PRE_DYING, or (op 1 of
                        // a program with ProgHdr::second_leaf_mat / PruneArgs::second_leaf_op set) by the item's prologue
    // partial updates (a finite-difference stencil row that moves one branch re-walks only the path from that branch to
    // the root, on tiles the base vector's walk left in HBM): see "save area" below
    F_POP_SAVED = 512,  // GEMM with F_POP_MUL (path programs): the saved product is save slot `slot`, not stack slot `slot`
    F_SAVE = 1024,      // GEMM (base vector's program): store this edge's tiles in save slots sv, sv+1, sv+2
    F_TABLE = 2048,     // OP_LOAD with F_SAVE (clade programs): the current tile, its exponents folded in, becomes 16 columns of the
                        // clade's table (a gather image of as many columns as the clade shows distinct sub-patterns), at matrix `mat`
    // work items of several pattern tiles (clade programs: the same short program over the clade's consecutive tiles).  The
    // program is cyclic: its tail requests what the next tile's head consumes, so tiles after the first have no prologue.
    F_NEXT_TILE = 4096  // on the last GEMM: its requests (F_PRE_EARLY: matrix pre_mat into the prefetch tile; F_PRE_DYING: pre_mat_b
                        // into the dying operand, the landing tile from then on) are the first leaf edges of the item's NEXT tile,
                        // read with that tile's codes, which the code queue holds by then (rows with ROW_NEXT_TILE), and `arg` is
                        // the FIRST product's matrix; all three only if another tile follows
};
constexpr int OP_FLAG_BITS = 13;   // DevOp::w0 = kind | flags << 2 | slot << (2 + OP_FLAG_BITS)
// A code-queue row (ProgHdr::*_leaf_row, DevOp::pre_row2/3) with this bit set is a row of the NEXT tile of the work item: read 64
// patterns further on, or not at all if the item has no further tile.
constexpr int32_t ROW_NEXT_TILE = 1 << 30;
// The host keeps one topology-only program (TreeOp, node ids) and resolves it per site class into the packed
// device form (DevOp, matrix indices) before upload, so the kernel never chases the (class, node) -> matrix table.
struct TreeOp {
    int32_t kind;    // OP_*
    int32_t node;    // child node whose branch this op crosses
    int32_t flags;   // F_*
    int32_t arg;     // LEAF: alignment row; GEMM: index of the next GEMM op (-1 none); CONST: 1 -> ones, 0 -> zeros
    int32_t slot;    // stack slot for F_PUSH / F_POP_MUL
    int32_t pre_node;    // F_PRE_EARLY/LATE: child node of the leaf edge whose columns are requested
    int32_t pre_node_b;  // F_PRE_DYING: child node of the second leaf edge requested by this op
    int32_t pre_row2;    // alignment rows whose codes refill the two-deep code queue after the request(s), -1 none:
    int32_t pre_row3;    //   the leaf edges two (and, with F_PRE_DYING, three) after the first one requested here
    int32_t sv;          // GEMM: first of this edge's three save slots; OP_LOAD: the save slot to read
};
struct alignas(16) DevOp {
    int32_t w0;        // kind | flags << 2 | slot << 14
    int32_t mat;       // matrix index of this edge in PruneArgs::P
    int32_t arg;       // LEAF: alignment row of the NEXT leaf edge (-1 none); GEMM: matrix of the next GEMM op (-1 none);
                       // CONST: 1 -> ones, 0 -> zeros
    int32_t pre_mat;   // F_PRE_EARLY/LATE: matrix of the leaf edge whose columns are requested
    int32_t pre_row2;  // code-queue refill rows (see TreeOp)
    int32_t pre_mat_b; // F_PRE_DYING: matrix of the second leaf edge
    int32_t pre_row3;
    int32_t sv;        // save slot (see TreeOp)
};
// Per-program header: programs of one launch may differ in length (path programs are a few ops long).
struct alignas(16) ProgHdr {
    int32_t n_ops;
    int32_t first_gemm_op, first_leaf_row, first_leaf_op, second_leaf_row, third_leaf_row;   // as in PruneArgs
    int32_t save_cls;  // class index inside the save area (the base vector's class with the same rate matrices)
    int32_t no_root;   // 1: a tree part -- the program ends by saving its tile (OP_LOAD with F_SAVE), no site likelihood is written
    int32_t alt_codes; // bit 0: a clade program -- its leaf rows are read from PruneArgs::codes_alt (the clade's distinct sub-patterns),
                       // in the slice of level alt_codes >> 2 (0 = innermost); bit 1: the program gathers from tables written in
                       // the SAME launch and waits for them first (PruneArgs::deps)
    int32_t first_gemm_mat, first_leaf_mat;   // matrices of ops first_gemm_op / first_leaf_op (-1 none): the item's prologue stages
                                              // the operand image and requests the columns without reading the ops first
    int32_t second_leaf_mat;   // >= 0: the program opens with two leaf edges of one node and this is the second one's matrix -- the
    int32_t fourth_leaf_row;   // prologue requests its columns into the landing tile; the code queue then continues at this row
    // A work item of a launch with an item list (PruneArgs::items) is a copy of its program's header with these three filled in:
    // one 64-byte scalar load stands between an item's number and everything its prologue needs.
    int32_t prog;              // the program (row of `ops`, of `hdr` and of out_f)
    int32_t tile0;             // first pattern tile
    int32_t item_tiles;        // consecutive tiles the item walks (> 1 only for programs with F_NEXT_TILE)
};
static_assert(sizeof(ProgHdr) == 64, "an item record is one s_load_dwordx16");
// Save area.  The base vector's walk stores, for every internal edge c (GEMM op with F_SAVE): slot sv = L_c, the operand
// of the edge's product; slot sv+1 = the parent's running product after this op (unless the op completes the parent);
// slot sv+2 = the edge's message P_c L_c before it is multiplied into the running product (if the op has F_POP_MUL).
// A tile is stored like a stack slot (8 KiB per wave) followed by its 64 per-lane exponents; slot s of class k,
// pattern tile t, wave w lives at ((s * save_K + k) * n_tiles + t) * 4 + w tiles of SAVE_TILE_BYTES.
constexpr size_t SAVE_TILE_BYTES = 16 * 64 * 8 + 64 * 4;
// Device programs are padded to a multiple of 64 ops plus one extra batch (the kernel reads one batch ahead).
inline int64_t dev_ops_padded(int64_t n_ops) { return ((n_ops + 63) / 64) * 64 + 64; }

// ---- kernel 1: batched P(t) = U diag(exp(lam t)) U^-1  (replaces matrix.rs:88-112) -----------
// For m in [0, M): eigensystem eig + eig_id[m]*EIG_STRIDE, time t_eff[m]; writes P + m*P_ELEMS in
// fragment order with row sums in column n_states.  eig_id[m] may carry BUILD_P_NO_OPERAND / BUILD_P_NO_GATHER: that image of
// matrix m is then left unwritten (nothing reads a leaf edge's operand image or an internal edge's gather image).
constexpr int32_t BUILD_P_NO_GATHER = 1 << 29, BUILD_P_NO_OPERAND = 1 << 30;
void launch_build_p(const double *eig, const int32_t *eig_id, const double *t_eff, double *P, int M, int n_states,
                    bool clamp_negative, hipStream_t stream, int32_t *zero = nullptr, int n_zero = 0);   // zero[0 .. n_zero): cleared
                                                                                   // by the launch's last workgroup (the pruning launches' counters)

// ---- kernel 0: batched spectral decomposition on the device  (replaces matrix.rs:31-75 where many matrices change per
// call: free-ratio fits, gene scans; the host solver of crab_eigen.cpp stays the default) ----------------------------
// For m in [0, M): S + m*NS*NS is the symmetrised rate matrix Pi^1/2 Q Pi^-1/2, padded with zeros to 64 x 64; the packed
// eigensystem (crab_eigen.h layout: U^T, U^-1, lambda) is written to pool + slot[m]*EIG_STRIDE.  One workgroup per
// matrix: parallel cyclic Jacobi in LDS, then one Newton-Schulz step on the eigenvectors.  sqrt_pi: [64], zero beyond n.
void launch_eigen(const double *S, const int32_t *slot, const double *sqrt_pi, double *pool, int M, int n_states, int *status,
                  hipStream_t stream);   // *status (zeroed by the caller) gets bit 0 set if a matrix did not converge

// ---- kernel 2: Felsenstein pruning, whole tree per (pattern tile, class)  ---------------------
// (replaces likelihood.rs:120-132,176-246 and branch_site_likelihood.rs:222-299)
struct PruneArgs {
    const DevOp *ops;         // [K][n_ops_pad] device programs, one per class
    int32_t n_ops;
    int32_t n_ops_pad;        // dev_ops_padded(n_ops)
    const double *P;          // all matrices of this evaluation
    int32_t K;
    const uint8_t *codes;     // [n_rows][n_pat_pad], values 0..n_states (n_states = missing)
    int64_t n_pat_pad;        // multiple of the tile width
    const double *pi_pad;     // [64], zero beyond n_states
    int32_t n_states;
    double *stack;            // global-memory spill area for saved products
    int32_t *stack_e;
    int32_t max_depth;
    double *out_f;            // [K][n_pat_pad]  scaled site likelihood per class
    int32_t *out_e;           // [K][n_pat_pad]  power-of-two exponent: f = out_f * 2^out_e
    const ProgHdr *hdr;       // [number of programs] per-program headers, or nullptr: n_ops and the five fields below apply to all
    const int32_t *cls_list;  // programs (= rows of out_f) this launch walks, or nullptr: 0 .. K-1
    int64_t xcd_bounds[9];    // with cls_list: work items [xcd_bounds[x], xcd_bounds[x + 1]) go to XCD x's workgroups (chunks of
                              // equal WORK, set by the host from the programs' lengths; the kernel's own cut is by item count)
    char *save;               // save area, or nullptr
    int32_t save_K;           // classes in the save area
    int32_t first_gemm_op;    // index of the first GEMM op (-1 if none)
    int32_t first_leaf_row;   // alignment row of the first leaf edge in program order (-1 if none)
    int32_t first_leaf_op;    // index of that op if it is op 0 (its columns are requested before the walk starts), else -1
    int32_t second_leaf_row;  // alignment rows of the second and third leaf edges (-1 if none)
    int32_t third_leaf_row;
    int32_t second_leaf_op;   // 1: ops 0 and 1 are leaf edges of one node, both requested by the prologue (op 1 into the landing
    int32_t fourth_leaf_row;  //    tile); the code queue then continues at this row.  -1: only op 0 is
    // Clade tables (subtree site repeats).  A clade whose leaves show U << n_pat distinct sub-patterns is walked once per
    // sub-pattern by a clade program (launch 1, codes_alt, work items from `items`); the message of its root's branch lands in a
    // table laid out like a gather image of U columns, and the main program reads it like a leaf edge: alignment rows >=
    // n_code_rows are rows of idx32 (per pattern: the column of its sub-pattern) instead of byte codes.
    const ProgHdr *items;     // [n_items] work items (ProgHdr::prog, tile0, item_tiles), or nullptr: the programs of cls_list in
                              // order, every tile, one tile per item
    int64_t n_items;
    const uint8_t *codes_alt; // [n_levels][n_code_rows][n_pat_pad] like codes, one slice per nesting level
    const uint32_t *idx32;    // [wide rows][n_pat_pad]
    int32_t n_code_rows;      // rows of `codes`; 0 = no wide rows in this launch
    int32_t *work_ctr;        // [8], zero before the launch: XCD x's workgroups take the items of its chunk in order, one at a
                              // time, whoever asks next (see prune_body); nullptr: a chunk's items are dealt round-robin
    // One launch for the clade programs of all levels (the engine's default): a program whose header has alt_codes bit 1 waits,
    // before its first gather, until every table it reads is complete.  A table and its counter are written and read by
    // workgroups of ONE XCD only: the XCDs' L2s are not coherent with each other inside a launch.  The host keeps a clade and
    // everything nested in it in one chunk of the item list, and with `deps` chunk x is handed out to the workgroups whose
    // hardware XCC id is x, wherever the device put them.
    const int32_t *deps;      // per program 16 ints [n, program x 6, tiles needed x 6, 0 0 0]; nullptr: not a merged clade launch
    int32_t *sync;            // zero before the evaluation.  sync[0]: set by a workgroup that gave up waiting (sync[1], [2]: see
                              // raise_fault); sync[4 + x]: the tiles the workgroups of XCD x have announced in all;
                              // sync[16 + p]: the tiles of program p whose table stores have reached the L2
    int32_t expect_tiles;     // > 0: this launch reads the tables of a merged clade launch that should have announced this many
                              // tiles in all; it writes NaN site likelihoods if the sync[4 .. 11] add up to something else
                              // or sync[0] is set
    int32_t *host_flag;       // with expect_tiles: pinned host memory as the device addresses it; set to 1 when the check fails, so
                              // that the host can return an error whatever the mixture's clamp makes of a NaN (nullptr: NaN only)
#ifdef CRAB_PROF
    unsigned long long *prof; // diagnostic builds: per-wave phase counters (set by launch_prune), or nullptr
#endif
};
// Stack slots the configured pruning kernel keeps in registers (the schedule must not park columns in the landing
// tile of a product that pops a deeper, global-memory slot), and whether it has a prefetch tile at all.
// Whether the device shows all eight XCDs (not a partition of the card): a merged clade launch has one chunk per XCD.
bool prune_device_has_all_xcds(int device);
int prune_reg_stack_slots();
bool prune_has_prefetch();
// Whether the configured kernel executes OP_LOAD / F_SAVE / F_POP_SAVED (the 64-pattern kernel does).
bool prune_supports_paths();
// Patterns per workgroup for the configuration launch_prune will use.
int prune_tile_width();
// Number of workgroups launch_prune will use for n_work = tiles*K work items on HIP device `device` (persistent grid).
int prune_grid_size(int64_t n_work, int device);
// Bytes of stack / stack_e needed for a grid of `grid` workgroups and `max_depth` slots.
size_t prune_stack_doubles(int grid, int max_depth);
size_t prune_stack_ints(int grid, int max_depth);
// Once per device, with that device current, before the first launch_prune on it: the pruning kernels use a little more than
// the 64 KiB of LDS a kernel gets without asking (two operand images + the work-item mailbox).  Returns a HIP error code.
int prune_configure_device();
void launch_prune(const PruneArgs &a, int grid, hipStream_t stream);

// ---- kernel 3: class mixture, log, weighted sum  ---------------------------------------------
// (replaces likelihood.rs:147-165,281-291,328-335,349-370 and branch_site_likelihood.rs:160-170)
enum : int32_t { MIX_LOGSUMEXP_CLAMP = 0, MIX_LOGSUMEXP_NOCLAMP = 1, MIX_LINEAR = 2 };
struct MixArgs {
    const double *f;         // [n_groups][K][n_pat_pad]
    const int32_t *e;
    const double *weights;   // [n_pat_pad] pattern multiplicities (0 on padding)
    const double *props;     // [n_groups][K] class proportions / frequencies
    int32_t K;               // classes per group
    int32_t n_groups;        // parameter vectors evaluated together (1 for a plain evaluation)
    int64_t n_pat;
    int64_t n_pat_pad;
    int32_t mode;
    double *pat_lnl;         // [n_groups][n_pat_pad] per-pattern log-likelihood
    double *partials;        // [n_groups][mix_partials(n_pat)]
    double *lnL;             // [n_groups]
};
int mix_blocks(int64_t n_pat);
int mix_partials(int64_t n_pat);   // partial sums per group the mixture kernels may write
void launch_mix(const MixArgs &a, hipStream_t stream);
// out[p*K + k] = ln max(f, 1e-300) (clamp != 0) or ln f
void launch_by_class(const double *f, const int32_t *e, int K, int64_t n_pat, int64_t n_pat_pad, int clamp, double *out,
                     hipStream_t stream);

// ---- kernel 4: grid integration of Bayes empirical Bayes site posteriors (analysis/beb.py:575-661) ----
// The K per-class site likelihoods are computed once; every grid point g is a mixture of n_mix of them,
// sum_c props[g][c] * F[cols[g][c]][p].  Stage 1 scales each pattern's K likelihoods by their maximum (beb.py:581-585),
// stage 2 gives ln f(X | g) per grid point (:589-607), the host turns those into posterior grid weights
// exp(lnfX_g - ln f(X)) (:609-612), stage 3 accumulates each pattern's class posteriors over the grid (:614-659).
constexpr int BEB_MAX_MIX = 16;
struct BebArgs {
    const double *f;          // [K][n_pat_pad] scaled likelihoods from the pruning kernel
    const int32_t *e;         //               and their exponents
    int32_t K;
    int64_t n_pat, n_pat_pad;
    const double *weights;    // [n_pat_pad] pattern multiplicities
    double *F;                // [K][n_pat_pad] work: likelihoods relative to the pattern's best class
    int32_t G, n_mix;
    const int32_t *cols;      // [G][n_mix]
    const double *props;      // [G][n_mix]
    double *lnfx;             // [G] out of stage 2
    const double *grid_w;     // [G] in to stage 3
    double *post;             // [n_pat][n_mix] out of stage 3
};
// Weighted class posteriors over a list of parameter vectors (analysis/beb.py:698-729): group b holds the K class
// likelihoods of parameter vector b; out[p][k] = sum_b w_b * p_bk f_bk(p) / max(sum_k p_bk f_bk(p), 1e-300) with
// f = max(f, 1e-300) as compute_site_log_likelihoods_by_class returns it.  Groups are added in order (deterministic).
constexpr int POST_MAX_K = 32;
void launch_class_posteriors(const double *f, const int32_t *e, int n_groups, int K, int64_t n_pat, int64_t n_pat_pad,
                             const double *props, const double *weights, double *out, hipStream_t stream);
void launch_beb_scale(const BebArgs &a, hipStream_t stream);
void launch_beb_grid_lnf(const BebArgs &a, hipStream_t stream);
void launch_beb_posteriors(const BebArgs &a, hipStream_t stream);

}  // namespace crab
