/*
 * crabml_amd.h -- C ABI of the MI355X codon-likelihood engine (libcrabml_amd.so).
 *
 * This is the drop-in boundary for crabML's native likelihood core: every entry
 * point below replaces one function of the PyO3 module `crabml.crabml_rust`
 * (/root/reference/rust/src/lib.rs:468-478) or a piece of state that module
 * rebuilds on every call.  Plain pointers and sizes only; no exception crosses
 * the ABI; every function returns a crab_status (0 = ok) and leaves a message
 * retrievable with crab_last_error().  All matrices are row-major double.
 *
 * Two layers:
 *   (1) a handle API (crab_create / crab_set_tree / crab_set_alignment /
 *       crab_eval_*), which keeps the tree program, the compressed site patterns
 *       and all workspaces resident in HBM between evaluations -- the reference
 *       boundary is stateless and re-uploads everything per call
 *       (lib.rs:162-187, likelihood.rs:97-106);
 *   (2) one-shot functions with the argument lists of the seven PyO3 functions,
 *       built on (1) with an internal handle cache keyed on the alignment buffer.
 *
 * Conventions shared by all calls
 *   tree      n_nodes entries (node_ids[k], parents[k]); parents[k] = -1 for "None".
 *             node_ids may be NULL (= 0..n_nodes-1).  Children keep their order of
 *             appearance; the root is the first node without a parent
 *             (likelihood.rs:36-81).  Any arity.
 *   leaves    leaf_node_ids[r] is the node that owns row r of the alignment
 *             (likelihood.rs:120-132).
 *   codes     int32 [n_leaves x n_sites] row-major; a code outside [0, n_states)
 *             is missing data (all-ones conditional likelihood).
 *   t         branch_lengths[v] = length of the branch above node v; the root's
 *             entry is ignored.
 *   n_states  <= 63 (61 for codons); padded to 64 on the device.
 */
#ifndef CRABML_AMD_H
#define CRABML_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crab_engine crab_engine;

typedef enum {
    CRAB_OK = 0,
    CRAB_ERR_ARG = 1,        /* shape / value errors: the reference raises ValueError or panics */
    CRAB_ERR_NO_ROOT = 2,    /* "No root node found" (likelihood.rs:57-59) */
    CRAB_ERR_EIGEN = 3,      /* "Eigendecomposition failed" (matrix.rs:54-55) */
    CRAB_ERR_DEVICE = 4,     /* HIP runtime error, no GPU, out of memory */
    CRAB_ERR_STATE = 5       /* call order: tree / alignment not set */
} crab_status;

/* flags for crab_eval_* */
#define CRAB_FLAG_NO_CLAMP 1u   /* drop the reference's max(f, 1e-300) clamps (likelihood.rs:161,332,356,450):
                                   exact log-likelihoods on trees long enough to underflow fp64 */

/* crab_eval_beb_grid over pattern slices (multi-GPU): f(X | grid point) is a product over all sites, so each rank first
 * calls with CRAB_FLAG_BEB_PARTIAL (lnfx receives this slice's sum_p w_p ln f_p(g); posteriors may be NULL), the ranks
 * add their lnfx vectors (one all-reduce of G doubles), and each calls again with CRAB_FLAG_BEB_GIVEN_LNFX (lnfx is now
 * input) to get the posteriors of its own sites. */
#define CRAB_FLAG_BEB_PARTIAL 2u
#define CRAB_FLAG_BEB_GIVEN_LNFX 4u

/* ---- lifetime ------------------------------------------------------------ */
/* device = HIP device ordinal (one engine per process/GPU is the intended use). */
int crab_create(int device, crab_engine **out);
/* One handle over several GPUs of this process (SURVEY 8(b) sketches crab_create(device_mask, ...); a list keeps the
 * order, which fixes the summation order).  The unchanged crabML is ONE Python process that calls the FFI once per
 * objective (core/likelihood_rust.py:226-288 -> lib.rs:146-191); with this handle that one call runs on every listed
 * GPU: crab_set_alignment compresses the alignment once and gives shard r a tile-aligned slice of the pattern list on
 * devices[r]; every crab_eval_* runs on all shards at once (one host thread and one stream per device, each rate
 * matrix decomposed once and shared), log-likelihoods (one, or B for batched calls) are added in shard order on the
 * host -- bit-identical from run to run -- and per-site outputs ([n_sites], [n_sites][K], BEB posteriors) are written by
 * the shard that owns the site.  The handle is used like any other engine with every function below; a device may be
 * listed more than once (several shards on one GPU).  n_devices in [1, 64]. */
int crab_create_multi(const int *devices, int n_devices, crab_engine **out);
/* Where CachedQMatrix::new (matrix.rs:31-75) runs.  0 (default): on the host, one task per matrix (0.25 ms for the 11
 * matrices of an M8 evaluation: the shortest latency when few matrices change per call).  1: on the device, one
 * workgroup per matrix (parallel cyclic Jacobi in registers + one re-orthogonalisation step, eigenvalues as Rayleigh
 * quotients; no host work beyond symmetrising the matrix, no synchronisation; 0.7-0.9 ms whatever the number of matrices
 * up to one per CU): for callers that change many rate matrices per call -- free-ratio fits (1 998 per
 * kappa row at 1000 taxa), gene scans with several engines per GPU (43 per gradient call each).  Results of the two
 * solvers agree to the accuracy of either (P entries within 2e-14); within one setting batched and single evaluations
 * stay bit-identical.  The environment variable CRAB_DEVICE_EIGEN=1 sets the default of new engines. */
int crab_set_eigensolver(crab_engine *e, int on_device);
/* Clade tables (subtree site repeats; on by default, CRAB_NO_CLADE_TABLES=1 switches them off for every engine).  The reference
 * computes every column at every node (likelihood.rs:176-195).  An evaluation (every vector of a batched call that walks the
 * whole tree, except the base vector of a batch with partial updates) instead walks each
 * chosen clade -- a subtree of at most 64 leaves whose leaves show few distinct sub-patterns; clades nest -- once per distinct
 * sub-pattern and lets whatever contains it gather the clade's message per (sub-)pattern; bit-identical to whole walks, a multiple fewer
 * products on alignments of related sequences.  crab_last_clade_tables: clades used by the last evaluation (0: whole walks)
 * and the sum of their distinct sub-pattern counts (over the shards of a multi-device handle). */
int crab_set_clade_tables(crab_engine *e, int on);
int crab_last_clade_tables(const crab_engine *e, int *n_clades, int64_t *sub_patterns);
/* Launches of the pruning kernel in the last evaluation: 1 (whole walks), 2 (partial updates, tree parts), or one per clade level
 * plus one (what a profiler's per-kernel average divides by). */
int crab_last_prune_launches(const crab_engine *e, int *n_launches);
/* Shards behind a handle: 1 for crab_create, n_devices for crab_create_multi. */
int crab_n_devices(const crab_engine *e);
void crab_destroy(crab_engine *e);
/* Message of the last failing call on `e` (or of the last failing call without a handle if e == NULL). */
const char *crab_last_error(const crab_engine *e);
/* 1 if a HIP device is usable, 0 otherwise (never faults). */
int crab_device_available(void);

/* ---- state: replaces Tree::from_structure (likelihood.rs:36-81) and the per-call
 *      leaf initialisation (likelihood.rs:120-132) -------------------------------- */
int crab_set_tree(crab_engine *e, int n_nodes, const int32_t *node_ids, const int32_t *parents,
                  const int32_t *leaf_node_ids, int n_leaves);
/* Uploads the alignment once.  compress != 0 collapses identical columns into weighted
 * site patterns (the reference computes every column; results are identical up to summation order).
 * pattern_begin/pattern_end select a slice of the (compressed) pattern list for multi-GPU sharding:
 * pass 0, -1 for all patterns. */
int crab_set_alignment(crab_engine *e, const int32_t *codes, int n_leaves, int64_t n_sites, int n_states,
                       int compress, int64_t pattern_begin, int64_t pattern_end);
int64_t crab_n_patterns(const crab_engine *e);       /* patterns held by this engine (after slicing) */
int64_t crab_n_patterns_total(const crab_engine *e); /* patterns of the whole alignment */
int64_t crab_n_sites(const crab_engine *e);

/* ---- evaluations ----------------------------------------------------------- */
/* compute_site_class_log_likelihood (lib.rs:146-191 -> likelihood.rs:256-292,349-370).
 * K = 1 with proportions = {1.0} is compute_log_likelihood (lib.rs:97-127 -> likelihood.rs:109-166).
 * Q: [K][n][n]; site_lnl: optional [n_sites] per-site log-likelihoods (NULL to skip;
 * only sites whose pattern lies in this engine's slice are written, others are left untouched). */
int crab_eval_site_classes(crab_engine *e, int K, const double *Q, const double *proportions, const double *pi,
                           const double *branch_lengths, unsigned flags, double *lnL, double *site_lnl);

/* B parameter vectors in one launch sequence: lnL[b] for b in [0, B), each as crab_eval_site_classes would give it
 * (bit-identical).  Q: [B][K][n][n]; proportions: [B][K]; branch_lengths: [B][n_nodes]; pi is shared.
 * Replaces the n_params+1 sequential FFI calls behind every finite-difference gradient of the reference's
 * optimisers (scipy L-BFGS-B without jac, optimize/optimizer.py:1197-1203 -> lib.rs:146-191 once per perturbation);
 * rate matrices that repeat within the batch (branch-length perturbations repeat all of them) are decomposed once. */
int crab_eval_site_classes_batch(crab_engine *e, int B, int K, const double *Q, const double *proportions,
                                 const double *pi, const double *branch_lengths, unsigned flags, double *lnL);

/* compute_site_log_likelihoods_by_class (lib.rs:214-258 -> likelihood.rs:303-338):
 * out[s*K + k] = ln max(f_{k,s}, 1e-300), shape [n_sites][K]. */
int crab_eval_site_lnl_by_class(crab_engine *e, int K, const double *Q, const double *pi,
                                const double *branch_lengths, unsigned flags, double *out);

/* Bayes empirical Bayes site posteriors on PAML's fixed grid: everything BEBCalculator.compute_site_posteriors_paml
 * (analysis/beb.py:513-661) does after building its rate matrices -- the K-class site likelihood table
 * (compute_site_log_likelihoods_by_class, beb.py:577 -> lib.rs:214-258), its per-site scaling (:581-585), ln f(X | g) for
 * every grid point (:589-607), ln f(X) (:609-611) and the class posteriors of every site (:614-659; in the reference
 * pure-Python loops over n_sites x 10 000 grid points, twice).  Grid point g mixes n_mix (<= 16) of the K classes:
 * f_s(g) = sum_c props[g*n_mix + c] * f_{cols[g*n_mix + c], s}.  posteriors: [n_sites][n_mix]; lnfx: optional [G];
 * ln_fx: optional scalar.  With flags = 0 the engine must hold the whole alignment; engines holding pattern slices use
 * the two-call protocol of CRAB_FLAG_BEB_PARTIAL / CRAB_FLAG_BEB_GIVEN_LNFX (above). */
int crab_eval_beb_grid(crab_engine *e, int K, const double *Q, const double *pi, const double *branch_lengths, int G,
                       int n_mix, const int32_t *cols, const double *props, unsigned flags, double *posteriors,
                       double *lnfx, double *ln_fx);

/* compute_branch_site_log_likelihood (lib.rs:287-345 -> branch_site_likelihood.rs:115-299).
 * Q3: [3][n][n] = Q(omega0), Q(1), Q(omega2), unnormalised; labels[n_labels], n_labels must be
 * n_nodes-1, indexed by node id, values 0/1.  lnL = sum_s ln sum_k freqs[k] f_{k,s}; no clamp. */
int crab_eval_branch_site(crab_engine *e, const double *Q3, double qfactor_back, double qfactor_fore,
                          double omega0, double omega2, const double *site_class_freqs, const double *pi,
                          const double *branch_lengths, const uint8_t *labels, int n_labels, unsigned flags,
                          double *lnL, double *site_lnl);

/* compute_log_likelihood_branch (lib.rs:363-430 -> likelihood.rs:385-455): one Q per node.
 * n_q must equal n_nodes and n_branch_lengths. */
int crab_eval_branch(crab_engine *e, int n_q, const double *Q, const double *pi, const double *branch_lengths,
                     int n_branch_lengths, unsigned flags, double *lnL, double *site_lnl);

/* ---- evaluations from model parameters: Q construction behind the boundary ---------------------
 * crabML builds every rate matrix in Python before each FFI call (models/codon.py:166-246 per matrix, called from the
 * site-class builders :350-376 (M1a), :441-461 (M2a), :518-536 (M3), :607-627 (M7), :709-729 (M8), :810-830 (M8a) ...,
 * from models/codon_branch.py:129-161 and from optimize/branch_site.py:150-170).  These entry points take the
 * parameters themselves, B parameter vectors per call (B = 1 for a single evaluation); 61-state alignments only.
 * All return lnL[b], b in [0, B); pi is shared by the batch; branch_lengths is [B][n_nodes]. */

/* Site-class mixtures M0 ... M9 once their class omegas and proportions are known (the beta / gamma quantile step,
 * a handful of scalar ppf calls, stays with the caller): Q_k = Q(kappa, omega_k, pi) / sum_k p_k mu_k, then exactly
 * crab_eval_site_classes_batch.  kappa: [B]; omegas, proportions: [B][K]. */
int crab_eval_codon_site_classes(crab_engine *e, int B, int K, const double *kappa, const double *omegas,
                                 const double *proportions, const double *pi, const double *branch_lengths,
                                 unsigned flags, double *lnL);
/* Branch (free-ratio / multi-ratio) model: omega_per_node [B][n_nodes] (the root's entry is built and ignored like the
 * reference's, codon_branch.py:154-161), each matrix normalised on its own; then as crab_eval_branch. */
int crab_eval_codon_branch(crab_engine *e, int B, const double *kappa, const double *omega_per_node, const double *pi,
                           const double *branch_lengths, unsigned flags, double *lnL);
/* Branch-site Model A from (kappa, omega0, omega2, p0, p1), each [B]: class frequencies
 * (codon_branch_site.py:108-161), Qfactors (:163-231), the three unnormalised matrices
 * (optimize/branch_site.py:163-165); then as crab_eval_branch_site.  p0 + p1 >= 1 or < 1e-10 is CRAB_ERR_ARG
 * (the reference raises ValueError). */
int crab_eval_codon_branch_site(crab_engine *e, int B, const double *kappa, const double *omega0, const double *omega2,
                                const double *p0, const double *p1, const double *pi, const double *branch_lengths,
                                const uint8_t *labels, int n_labels, unsigned flags, double *lnL);
/* The rate matrices crab_eval_codon_site_classes builds for one parameter vector, [K][61][61].  Host only. */
int crab_build_site_class_q_matrices(int K, double kappa, const double *omegas, const double *proportions,
                                     const double *pi, double *Q_out);

/* Class posteriors of every site averaged over a weighted list of parameter vectors: what
 * BEBCalculator.compute_site_posteriors (analysis/beb.py:663-731, the adaptive-grid path) accumulates in Python, one
 * compute_site_log_likelihoods call and one [n_sites, K] table per grid point.  For vector b: lik_k = max(f_k, 1e-300) p_bk
 * (:718-721), normalised over the classes with the row sum floored at 1e-300 (:724-726), added with weight w_b (:729).
 * kappa, weights: [B]; omegas, proportions: [B][K] (K <= 32); branch_lengths: [n_nodes], shared (the reference fixes
 * them at the MLE); posteriors: [n_sites][K].  Call it once per batch of grid points and add the results. */
int crab_eval_codon_class_posteriors(crab_engine *e, int B, int K, const double *kappa, const double *omegas,
                                     const double *proportions, const double *weights, const double *pi,
                                     const double *branch_lengths, unsigned flags, double *posteriors);
/* The quantile step of the discrete-distribution models: omegas_out[j] = ppf((2j+1)/(2K)), j in [0, K), of
 * beta(a, b) (kind 0: M7 / M8 / M8a, models/codon.py:597-604, 694-701, 795-802) or of gamma(shape a, rate b)
 * (kind 1: M5, :892-899), shape parameters floored at 0.005 like the reference (:570-571, :864-865).  Replaces the
 * scipy.stats ppf calls (about 1 ms per objective call in the reference).  Host only. */
int crab_discretise_omega(int kind, double a, double b, int K, double *omegas_out);
/* The same step for the two models whose classes discretise a MIXTURE of two distributions, M6 and M9 (models/codon.py:
 * 982-1020 and 1183-1222): omegas_out[j] is the root of cdf(x) = (2j+1)/(2K) on [1e-4, 99], found by Brent's method as
 * scipy.optimize.brentq runs it with its default tolerances (what the reference calls), or the reference's fallback 10 p
 * where the CDF does not change sign over that bracket.  kind 2 (M6): params = {p0, alpha1, beta1, alpha2}, cdf = p0
 * Gamma(alpha1, rate beta1) + (1 - p0) Gamma(alpha2, rate alpha2); kind 3 (M9): params = {p0, p_beta, q_beta, alpha,
 * beta_gamma}, cdf = p0 Beta(p_beta, q_beta) + (1 - p0) Gamma(alpha, rate beta_gamma); parameters clamped like the
 * reference's constructors.  Host only. */
int crab_discretise_omega_mixture(int kind, const double *params, int n_params, int K, double *omegas_out);
/* Model A's class frequencies [4] and Qfactors {background, foreground} as crab_eval_codon_branch_site computes them
 * (codon_branch_site.py:108-231).  Host only. */
int crab_model_a_quantities(double kappa, double omega0, double omega2, double p0, double p1, const double *pi,
                            double *site_class_freqs, double *qfactors);

/* Timing of the last crab_eval_* on this engine, in milliseconds (HIP events on the engine's
 * stream): whole device section, and the pruning kernel alone.  For bench.py's roofline block. */
int crab_last_timing(const crab_engine *e, double *device_ms, double *prune_kernel_ms, double *host_eigen_ms);
/* Static facts about the pruning launches of the last evaluation: number of dense
 * (pattern x branch x class) updates executed as MFMA products, and of leaf-edge updates done as gathers
 * (with partial updates: those actually executed, not those a whole walk of every vector would take). */
int crab_last_work(const crab_engine *e, int64_t *dense_updates, int64_t *gather_updates);

/* Partial updates (second half).  In every batched call (B > 1) a vector whose transition matrices differ
 * from vector 0's on exactly ONE branch -- a finite-difference row that moves one branch length, or one branch's omega
 * under the free-ratio model -- is evaluated by re-walking only the path from that branch to the root, on conditional
 * likelihoods that vector 0's walk left in HBM (three 8 KiB tiles per internal edge, wave and class; used when at
 * least 4 vectors qualify and the area fits CRAB_PATH_SAVE_GB, default 96).  Same products in the same order as a
 * whole walk: results are bit-identical (tested); CRAB_NO_PATH_UPDATES=1 switches it off.  Put the centre of a stencil
 * first.  This call reports how many vectors of the last evaluation took the short path.  Replaces the n_params + 1
 * whole-tree evaluations per gradient of optimize/optimizer.py:1197-1203. */
int crab_last_partial_updates(const crab_engine *e, int *n_vectors);

/* Tree parts (a finer work quantum).  A work item of the pruning kernel is one workgroup walking the whole tree for 64
 * patterns of one class; a single evaluation with fewer than 3.2 items per resident workgroup (one GPU's share of a
 * sharded alignment: 392 items for 512 workgroups at BASELINE config 3 over 8 GPUs) is cut below a set of internal nodes
 * into subtree programs, which run as items of their own and leave their conditional likelihoods in HBM, and a root
 * program that picks them up in a second launch.  Same ops in the same order: bit-identical results (tested);
 * CRAB_NO_TREE_PARTS=1 switches it off.  Reports the number of subtree programs of the last evaluation (0: uncut). */
int crab_last_tree_parts(const crab_engine *e, int *n_parts);

/* Eigensystem pool of this engine since creation: distinct decompositions found resident (hits), decomposed anew
 * (misses), and entries resident now.  Replaces nothing in the reference (it decomposes every matrix on every call,
 * lib.rs:163-169,387-393); lets callers and tests see that a parameter change decomposed only what it had to. */
int crab_eig_pool_stats(const crab_engine *e, int64_t *hits, int64_t *misses, int64_t *resident);

/* Copies out the transition matrices the last crab_eval_* built on the device for class k:
 * P_out[v][i][j] = P(branch above node v)[i][j], shape [n_nodes][n][n].  Lets a caller (and the parity
 * tests) check kernel 1 and kernel 2 separately.  Replaces nothing in the reference: its P matrices
 * (matrix.rs:108-112) never leave the call. */
int crab_export_transition_matrices(crab_engine *e, int k, double *P_out);

/* ---- stateless utilities ----------------------------------------------------- */
/* matrix_exponential (lib.rs:61-79 -> matrix.rs:31-102): P = exp(Q t), computed on the device
 * with the same batched kernel the evaluations use.  P_out: [n][n]. */
int crab_matrix_exponential(int device, int n, const double *Q, const double *pi, double t, double *P_out);
/* crab_matrix_exponential for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc, code that runs between the capture sections of a HIP graph (the call itself copies synchronously and launches
 * on the null stream, so it must not be made WHILE a stream capture is active; it restores the caller's current device).  It has no counterpart in the reference (whose expm allocates a fresh ndarray
 * per call, matrix.rs:88-102).  Same arguments and results as crab_matrix_exponential (bit-identical: same host
 * decomposition, same kernel); the eigensystem, time and output images that crab_matrix_exponential stages in
 * device buffers allocated per call live in a file-scope device array instead -- 131 616 bytes per GPU, held from the
 * first call on that GPU until the process ends.  Calls no hipMalloc / hipFree / hipHostMalloc.  Limit: n <= 63 (one
 * 64-state eigensystem and one transition matrix); larger n is CRAB_ERR_ARG.  Every call on a GPU shares that GPU's
 * array: calls from several threads of one process are serialised inside the library; the array keeps the last
 * call's contents, each call overwrites everything its kernel reads. */
int crab_matrix_exponential_noalloc(int device, int n, const double *Q, const double *pi, double t, double *P_out);
/* build_codon_q_matrix_py (lib.rs:445-463 -> qmatrix.rs:22-80).  Host only. */
int crab_build_codon_q_matrix(double kappa, double omega, const double *pi, int has_normalization_factor,
                              double normalization_factor, double *Q_out);

/* ---- host-only introspection: the pieces of host logic, callable without a GPU ---- */
/* Tree::from_structure + post-order (likelihood.rs:36-81) compiled into the device program.
 * ops_out receives 5 int32 per op {kind, node, flags, arg, slot} (see crab_kernels.h);
 * pass ops_out = NULL to query n_ops first. */
int crab_compile_tree(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids,
                      int n_leaves, int32_t *ops_out, int ops_capacity, int *n_ops, int *max_depth);
/* The program of a partial update (crab_last_partial_updates): what is recomputed when only the branch above `node`
 * moves, 6 int32 per op {kind, node, flags, arg, slot, save slot}; node = -1 gives the whole-tree program in the same
 * 6-field form.  n_save_slots: tiles per (class, pattern tile, wave) the base walk saves. */
int crab_compile_path_program(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids,
                              int n_leaves, int node, int32_t *ops_out, int ops_capacity, int *n_ops, int *n_save_slots);
/* The programs of a tree cut into parts of at most `target` products (crab_last_tree_parts): part = -1 gives the root
 * program, part = i the i-th subtree program, in the 6-field form of crab_compile_path_program. */
int crab_compile_tree_part(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                           int target, int part, int32_t *ops_out, int ops_capacity, int *n_ops, int *n_parts);
/* Clade tables as programs, for tests (no device needed; `codes` holds the patterns as given, [n_leaves][n_patterns]).  part = -1:
 * the main program, in which each outermost clade is one leaf-edge op whose alignment row is n_leaves + its wide row; part = j >= 0
 * (clades come by level, inner ones first): clade j's program (its subtree with the clades inside it as such leaf-edge ops, the
 * product on its root's branch, a store with flags F_SAVE | F_TABLE), its root node, the clade that gathers from it
 * (parent_clade, -1: the main program), its wide row there, its number of distinct sub-patterns U, its index for every pattern
 * of its consumer (index_out: n_patterns entries for the main program, U of the parent clade otherwise; capacity n_patterns) and
 * its sub-patterns' codes (sub_codes_out [n_leaves][U], n_states on rows outside the clade).  Ops as in crab_compile_tree_part. */
int crab_compile_clade_tables(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                              const int32_t *codes, int64_t n_patterns, int n_states, int part, int32_t *ops_out, int ops_capacity,
                              int *n_ops, int *n_clades, int *clade_root, int *parent_clade, int *wide_row, int64_t *n_sub_patterns,
                              uint32_t *index_out, int32_t *sub_codes_out);
/* What the clade-table plan of a set of patterns executes, per site class (host only; for tests and for choosing how an
 * alignment is cut into shards): the clades tabled, the sum of their distinct sub-patterns, the products of one evaluation
 * (clade programs over padded sub-patterns + main program per pattern) and the products of whole walks. */
int crab_plan_clade_work(int n_nodes, const int32_t *node_ids, const int32_t *parents, const int32_t *leaf_node_ids, int n_leaves,
                         const int32_t *codes, int64_t n_patterns, int n_states, int *n_clades, int64_t *sub_patterns,
                         int64_t *products, int64_t *whole_walk_products);
/* Site-pattern compression as crab_set_alignment(compress=1) performs it. Outputs may be NULL. */
int crab_compress_patterns(const int32_t *codes, int n_leaves, int64_t n_sites, int n_states,
                           int64_t *site_to_pattern, double *pattern_weight, int64_t *first_site,
                           int64_t *n_patterns);
/* CachedQMatrix::new (matrix.rs:31-75) as the engine does it on the host: U, lam, U^-1 (n x n, n, n x n). */
int crab_spectral_decompose(int n, const double *Q, const double *pi, double *U, double *lam, double *Uinv);

/* Host worker pool under concurrent callers (several engines, one host thread each): `callers` threads run `rounds`
 * parallel loops of n indices at once and check that every index ran exactly once. */
int crab_host_pool_selftest(int callers, int n, int rounds);

/* ---- one-shot forms with the argument lists of the PyO3 functions ------------- */
/* Each looks up (or creates) a cached engine for (codes pointer, shape, checksum, tree) on `device`.
 * device = -1: the GPUs listed in the environment variable CRAB_DEVICES (comma-separated ordinals, e.g. "0,1,2,3,4,5,6,7";
 * unset: device 0) through crab_create_multi, so that one unchanged crabML process uses the whole node. */
int crab_compute_log_likelihood(int device, int n, const double *q, const double *pi, int n_nodes,
                                const int32_t *node_ids, const int32_t *parents, const double *branch_lengths,
                                const int32_t *codes, int n_leaves, int64_t n_sites, const int32_t *leaf_node_ids,
                                double *lnL);
int crab_compute_site_class_log_likelihood(int device, int K, int n, const double *q_matrices,
                                           const double *proportions, const double *pi, int n_nodes,
                                           const int32_t *node_ids, const int32_t *parents,
                                           const double *branch_lengths, const int32_t *codes, int n_leaves,
                                           int64_t n_sites, const int32_t *leaf_node_ids, double *lnL);
int crab_compute_site_log_likelihoods_by_class(int device, int K, int n, const double *q_matrices, const double *pi,
                                               int n_nodes, const int32_t *node_ids, const int32_t *parents,
                                               const double *branch_lengths, const int32_t *codes, int n_leaves,
                                               int64_t n_sites, const int32_t *leaf_node_ids, double *out);
int crab_compute_branch_site_log_likelihood(int device, int n, const double *q_omega0, const double *q_omega1,
                                            const double *q_omega2, double qfactor_back, double qfactor_fore,
                                            double omega0, double omega2, const double *site_class_freqs,
                                            const double *pi, int n_nodes, const int32_t *node_ids,
                                            const int32_t *parents, const double *branch_lengths,
                                            const uint8_t *branch_labels, int n_labels, const int32_t *codes,
                                            int n_leaves, int64_t n_sites, const int32_t *leaf_node_ids, double *lnL);
int crab_compute_log_likelihood_branch(int device, int n, const int32_t *codes, int n_leaves, int64_t n_sites,
                                       const int32_t *parent_indices, int n_nodes, const int32_t *leaf_node_ids,
                                       const double *Q_matrices, int n_q, const double *branch_lengths,
                                       int n_branch_lengths, const double *pi, double *lnL);
/* Drops every cached one-shot engine (frees their HBM). */
void crab_clear_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* CRABML_AMD_H */
