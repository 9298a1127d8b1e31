// gfx950 (MI355X / CDNA4) kernels of the codon-likelihood engine.
//
//   build_p_kernel      batched P(t) = U diag(exp(lam t)) U^-1 on fp64 MFMA, written in the two images of crab_kernels.h
//   prune_kernel_nt1/2  Felsenstein pruning: one workgroup = 64*NT site patterns of one site class, each wave walks
//                       the whole tree program for its 16*NT patterns.  Conditional likelihoods live in fixed
//                       registers outside the compiler's allocation; every internal edge is a 64x64 fp64 MFMA
//                       product whose accumulator tile is the next product's B operand; leaf edges are column
//                       gathers prefetched under the products; sibling products are saved in register tiles;
//                       power-of-two rescaling is fused; P operand images stream through LDS by LDS-DMA
//   mix_kernel          per-pattern class mixture, log, pattern-weighted deterministic sum (per parameter vector)
//   mix_wide_kernel     the same for log-sum-exp mixtures of up to 16 classes, one (pattern, class) per lane (the default there)
//   mix_final_kernel    mixture and final sum in one launch when the patterns fit one block of 256
//
// fp64 MFMA shape: the pruning products use v_mfma_f64_4x4x4_4b_f64.  Measured on MI355X (scripts/micro/,
// profiles/r1_micro_*): it sustains 73-76 TFLOP/s against 35-49 TFLOP/s for v_mfma_f64_16x16x4_f64, and its
// D lane map equals its B lane map, so products chain through registers.
// Written for gfx950 wave64 only.  No other architecture is supported.
#include "crab_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <utility>

namespace crab {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
// Host-written tables the kernels only read (item records, headers, program lists), through the constant address space: a
// wave-uniform read is then a scalar load whatever the alias analysis makes of the kernel's own stores and atomics.
#define CONST_PTR(T, p) ((const __attribute__((address_space(4))) T *)(unsigned long long)(p))


// =============================================================================================
// kernel 1: batched transition matrices
// =============================================================================================
// One workgroup (4 waves) builds `per_wg` consecutive matrices of the list; wave w produces rows i in [16w, 16w+16) of P.
// It computes the transposed tiles D^T[j][i] = sum_k Uinv[k][j] * (U[i][k] e^{lam_k t}) with
// A = Uinv^T (M index = j), B = (U diag)^T (N index = i) on v_mfma_f64_16x16x4_f64: with the f64 C/D map
// (col = lane&15, row = (lane>>4) + 4*reg) lane l / register r of tile jt holds
// P[16w + (l&15)][16jt + 4r + (l>>4)], stored into both images of crab_kernels.h with 16-byte stores.
// The A operand (64 scattered 8-byte loads per lane, two thirds of the kernel's loads) depends on the eigensystem only and
// stays in registers while consecutive matrices share one -- all branches of a site class do.  The 64 exponentials of a
// matrix are computed by one wave and shared through LDS (every lane used to compute the 16 it needs: 16-fold redundant,
// and at ~250 cycles per fp64 exp more work than the kernel's MFMAs).
__global__ __launch_bounds__(256, 2) void build_p_kernel(const double *__restrict__ eig, const int32_t *__restrict__ eig_id,
                                                      const double *__restrict__ t_eff, double *__restrict__ P, int n_states,
                                                      int clamp_negative, int M, int per_wg, int32_t *__restrict__ zero, int n_zero)
{
    if (zero && blockIdx.x == gridDim.x - 1)
        for (int z = threadIdx.x; z < n_zero; z += 256) zero[z] = 0;
    const int lane = threadIdx.x & 63;
    const int mt = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int ns_jt = n_states >> 4, ns_r = (n_states >> 2) & 3, ns_g = n_states & 3;
    const int i = 16 * mt + c;
    const bool hi_half = (c & 4) != 0;
    const int m0 = blockIdx.x * per_wg, m1 = m0 + per_wg < M ? m0 + per_wg : M;
    __shared__ double ex[2][NS];   // e^{lam_k t} of the current matrix: 64 values, computed once (not once per lane that uses them)
    double a[16][4];
    int cur = -1;
    for (int m = m0; m < m1; ++m) {
        // (bits 29 and 30 of the entry: this matrix's gather / operand image has no reader -- a leaf edge is only ever
        // gathered from, an internal edge only ever multiplied with -- and is not written)
        const int raw_id = __builtin_amdgcn_readfirstlane(eig_id[m]);
        const int id = raw_id & (BUILD_P_NO_GATHER - 1);
        const bool store_operand = !(raw_id & BUILD_P_NO_OPERAND), store_gather = !(raw_id & BUILD_P_NO_GATHER);
        const double *E = eig + (size_t)id * EIG_STRIDE;
        const double *UT = E + EIG_UT, *Uinv = E + EIG_UINV, *lam = E + EIG_LAM;
        if (id != cur) {
            cur = id;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) a[s][jt] = Uinv[(4 * s + g) * NS + 16 * jt + c];
        }
        const double t = t_eff[m];
        if (threadIdx.x < NS) ex[m & 1][threadIdx.x] = exp(lam[threadIdx.x] * t);
        __syncthreads();   // (two buffers: the readers of this one, two matrices ago, have passed the barrier in between)

        v4d acc[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 4 * s + g;
            const double b = UT[k * NS + 16 * mt + c] * ex[m & 1][k];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][jt], b, acc[jt], 0, 0, 0);
        }
        // exp(Qt) >= 0; the eigen route leaves entries of -1e-15 where the true value is (near) zero, e.g. everywhere off
        // the diagonal at t = 0.  The pruning pass rescales each pattern by its largest *positive* entry, so data that such
        // a matrix makes impossible would otherwise blow negative noise up into infinities: evaluations clamp it to zero
        // (as PAML does); crab_matrix_exponential returns the raw values like the reference (matrix.rs:88-102).
        if (clamp_negative) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[jt][r] = acc[jt][r] < 0.0 ? 0.0 : acc[jt][r];
        }
        // row sums over the real states (padded columns are exact zeros)
        double rs = 0.0;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) rs += acc[jt][r];
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        // column n_states (a padding column, zero so far) receives the row sum
        double *out = P + (size_t)m * P_ELEMS;
        // 16-byte stores.  Operand image: registers r, r+1 (r even) of a lane are neighbours there (the (j>>2)&1 bit).
        // Gather image: rows i and i^4 are neighbours (the (i>>2)&1 bit) and live in lanes c and c^4, so each pair of lanes
        // swaps one value per register pair: the lane with bit 2 of c clear stores column j(r), the other column j(r+1).
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                double v0 = acc[jt][r], v1 = acc[jt][r + 1];
                if (jt == ns_jt && r == ns_r && g == ns_g) v0 = rs;
                if (jt == ns_jt && r + 1 == ns_r && g == ns_g) v1 = rs;
                const int j0 = 16 * jt + 4 * r + g, j1 = j0 + 4;
                if (store_operand) *(v2d *)(out + p_operand_index(i, j0)) = (v2d){v0, v1};
                // what the partner lane (c ^ 4) needs from this lane: its column is j1 if this lane is the low half, j0 otherwise
                if (store_gather) {
                    const double give = hi_half ? v0 : v1;
                    const double got = __shfl_xor(give, 4);
                    const v2d pair = hi_half ? (v2d){got, v1} : (v2d){v0, got};
                    *(v2d *)(out + p_gather_index(i & ~4, hi_half ? j1 : j0)) = pair;
                }
            }
    }
}

void launch_build_p(const double *eig, const int32_t *eig_id, const double *t_eff, double *P, int M, int n_states,
                    bool clamp_negative, hipStream_t stream, int32_t *zero, int n_zero)
{
    if (M <= 0) {
        if (zero && n_zero > 0) (void)hipMemsetAsync(zero, 0, (size_t)n_zero * sizeof(int32_t), stream);
        return;
    }
    // one round of workgroups (two fit a CU at this kernel's register count), each with a run of the matrix list
    int per_wg = (M + 511) / 512;
    static const int per_wg_env = std::getenv("CRAB_BUILD_P_PER_WG") ? std::atoi(std::getenv("CRAB_BUILD_P_PER_WG")) : 0;   // experiments
    if (per_wg_env > 0) per_wg = per_wg_env;
    const int grid = (M + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(build_p_kernel, dim3(grid), dim3(256), 0, stream, eig, eig_id, t_eff, P, n_states, clamp_negative ? 1 : 0, M, per_wg, zero, n_zero);
}

// =============================================================================================
// kernel 0: batched symmetric eigensolver (parallel cyclic Jacobi, matrix and vectors in registers)
// =============================================================================================
// One workgroup (256 threads) per matrix.  The 64 indices (n_states real ones, the rest padding with exact zero rows
// and columns) sit in 64 *seats*: seat k and seat 32 + k (k < 32) form pair k.  A step rotates all 32 pairs at once,
// A <- J^T A J and V <- V J, then every index but the one in seat 0 moves one seat on along the ring
// 1, 2, .., 31, 63, 62, .., 32 (the circle method of round-robin tournaments): 63 steps = one sweep let every pair meet
// once and put every index back into its first seat.
//   - A in seat order: thread (a, cq) holds rows {a, 32 + a} x columns {4cq .. 4cq+3, 32+4cq .. 32+4cq+3}: four 2 x 2
//     blocks, each rotated from both sides in registers.  The thread whose patch holds the diagonal block of pair a
//     computes that pair's rotation and posts (c, s) in a table in LDS.  The move is the only traffic: the patch is
//     written to LDS at its seats (eight 16-byte stores), and read back from the seats that move into the patch's
//     (sixteen 8-byte loads at constant offsets).
//   - V: the same patch of V (rows a and 32 + a, which stay); columns rotate with the patch's four (c, s) and move by
//     renaming, the four entries that cross to the neighbouring thread by lane shuffles.
// Two barriers per step.  Sweeps stop when one passes without a rotation (off-diagonal entries below 2^-53 of the
// geometric mean of their diagonal entries), at most 24.  Then one Newton-Schulz step V <- V (3 I - V^T V) / 2
// (crab_eigen.cpp:reorthogonalise has the reason), eigenvalues as Rayleigh quotients v^T S v of the corrected
// vectors (the rotated diagonal carries the rounding of ~500 rotations per index), and the result packed for kernel 1.
constexpr int EJ_LD = 66;      // LDS row stride in doubles: even (16-byte pieces stay aligned), consecutive rows 16 B apart in the banks
constexpr size_t EJ_LDS_BYTES = (size_t)(2 * NS * EJ_LD + 64 + NS) * sizeof(double) + 16;

__device__ __forceinline__ int ej_src(int s)                // the seat whose index moves into seat s after a step
{
    return s == 0 ? 0 : s == 1 ? 32 : s < 32 ? s - 1 : s < 63 ? s + 1 : 31;
}

// sqrt(z) and 1 / sqrt(z) for a positive, normal z: the hardware's estimate and two coupled Goldschmidt steps (fma only;
// ocml's sqrt/rsqrt carry range scaling that the rotation, on the critical path of every step, does not need)
__device__ __forceinline__ void ej_sqrt_rsqrt(double z, double &root, double &inv_root)
{
    const double y = __builtin_amdgcn_rsq(z);
    double h = z * y, g = 0.5 * y;
    double r = __builtin_fma(-g, h, 0.5);
    h = __builtin_fma(h, r, h), g = __builtin_fma(g, r, g);
    r = __builtin_fma(-g, h, 0.5);
    h = __builtin_fma(h, r, h), g = __builtin_fma(g, r, g);
    root = h, inv_root = 2.0 * g;
}

__device__ __forceinline__ double ej_sel(const double (&x)[4], int m)
{
    return m == 0 ? x[0] : m == 1 ? x[1] : m == 2 ? x[2] : x[3];
}

__global__ __launch_bounds__(256) void eigen_kernel(const double *__restrict__ S, const int32_t *__restrict__ slot,
                                                    const double *__restrict__ sqrt_pi, double *__restrict__ pool, int n,
                                                    int max_sweeps, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *X = reinterpret_cast<double *>(smem_raw);           // [64][EJ_LD] A on the move; scratch of the epilogue
    double *Vs = X + NS * EJ_LD;                                // [64][EJ_LD] V, once the sweeps are done
    double *tab = Vs + NS * EJ_LD;                              // [32] (c, s) of the step's rotations
    double *lam = tab + 64;                                     // [64]
    int *n_rot = reinterpret_cast<int *>(lam + NS);             // rotations made in the current sweep
    const int t = threadIdx.x;
    const double *Sm = S + (size_t)blockIdx.x * NS * NS;        // 64 x 64, zero beyond n (the host pads)

    const int a = t >> 3, cq = t & 7;                           // A: pair of rows, eighth of the columns
    const bool diag = cq == (a >> 2);
    const int md = a & 3;                                       // where the diagonal block sits in the patch
    double xtt[4], xtb[4], xbt[4], xbb[4];                      // A: rows {a, 32 + a} x seats {4cq + m, 32 + 4cq + m}
    double vtt[4], vtb[4], vbt[4], vbb[4];                      // V: the same patch (its rows do not move)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        xtt[m] = Sm[a * NS + 4 * cq + m], xtb[m] = Sm[a * NS + 32 + 4 * cq + m];
        xbt[m] = Sm[(32 + a) * NS + 4 * cq + m], xbb[m] = Sm[(32 + a) * NS + 32 + 4 * cq + m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        vtt[m] = a == 4 * cq + m ? 1.0 : 0.0, vtb[m] = 0.0;
        vbt[m] = 0.0, vbb[m] = a == 4 * cq + m ? 1.0 : 0.0;
    }
    const int rt = ej_src(a) * EJ_LD, rb = ej_src(32 + a) * EJ_LD;   // where the patch's next contents are, after a store at the seats
    int ct[4], cb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) ct[m] = ej_src(4 * cq + m), cb[m] = ej_src(32 + 4 * cq + m);
    if (t == 0) *n_rot = 0;
    __syncthreads();

#ifdef EJ_PROF
    long long pf[6] = {0, 0, 0, 0, 0, 0}, pf_t = clock64();
    int pf_sweeps = 0;
#define EJ_MARK(i) { const long long now_ = clock64(); pf[i] += now_ - pf_t; pf_t = now_; }
#else
#define EJ_MARK(i)
#endif
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        int made_here = 0;
#ifdef EJ_PROF
        ++pf_sweeps;
#endif
        for (int step = 0; step < 63; ++step) {
            EJ_MARK(5)
            bool clear = false;                                 // the pair's off-diagonal entries become exact zeros
            if (diag) {
                double c = 1.0, sn = 0.0;
                // (a pair with a padding index has a_pq = 0 exactly and is left alone)
                const double app = ej_sel(xtt, md), aqq = ej_sel(xbb, md), apq = ej_sel(xtb, md);
                if (apq * apq > 1.232595164407831e-32 * fabs(app * aqq)) {          // |a_pq| > 2^-53 sqrt(|a_pp a_qq|)
                    // t = sign(d) b / (|d| + sqrt(d^2 + b^2)) with d = a_qq - a_pp, b = 2 a_pq; c = 1 / sqrt(1 + t^2), s = t c
                    const double d = aqq - app, b2 = 2.0 * apq;
                    double root, inv;
                    ej_sqrt_rsqrt(d * d + b2 * b2, root, inv);
                    const double big = fabs(d) + root;
                    ej_sqrt_rsqrt(big * big + b2 * b2, root, inv);
                    c = big * inv;
                    sn = (d >= 0.0 ? b2 : -b2) * inv;
                    // a rotation is "quiet" if both the entry and the angle are below 2^-32 (relative): see the end of the sweep
                    made_here += (apq * apq > 5.421010862427522e-20 * fabs(app * aqq) || fabs(sn) > 2.3283064365386963e-10) ? 0x10001 : 1;
                    clear = true;
                } else if (apq != 0.0) {
                    clear = true;                               // negligible: drop it (both copies) so that it cannot creep back
                }
                *reinterpret_cast<v2d *>(tab + 2 * a) = (v2d){c, sn};
            }
            EJ_MARK(0)
            __syncthreads();
            EJ_MARK(1)
            {
                const v2d ra = *reinterpret_cast<const v2d *>(tab + 2 * a);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const v2d rc = *reinterpret_cast<const v2d *>(tab + 2 * (4 * cq + m));
                    // columns:  (x_p, x_q) <- (c x_p - s x_q, s x_p + c x_q);  then rows, the same way
                    const double tt = rc.x * xtt[m] - rc.y * xtb[m], tb = rc.y * xtt[m] + rc.x * xtb[m];
                    const double bt = rc.x * xbt[m] - rc.y * xbb[m], bb = rc.y * xbt[m] + rc.x * xbb[m];
                    xtt[m] = ra.x * tt - ra.y * bt, xbt[m] = ra.y * tt + ra.x * bt;
                    xtb[m] = ra.x * tb - ra.y * bb, xbb[m] = ra.y * tb + ra.x * bb;
                    if (clear && m == md) xtb[m] = 0.0, xbt[m] = 0.0;
                    const double v0 = rc.x * vtt[m] - rc.y * vtb[m], v1 = rc.y * vtt[m] + rc.x * vtb[m];
                    const double v2 = rc.x * vbt[m] - rc.y * vbb[m], v3 = rc.y * vbt[m] + rc.x * vbb[m];
                    vtt[m] = v0, vtb[m] = v1, vbt[m] = v2, vbb[m] = v3;
                }
                double *row_t = X + a * EJ_LD + 4 * cq, *row_b = X + (32 + a) * EJ_LD + 4 * cq;
                *reinterpret_cast<v2d *>(row_t) = (v2d){xtt[0], xtt[1]};
                *reinterpret_cast<v2d *>(row_t + 2) = (v2d){xtt[2], xtt[3]};
                *reinterpret_cast<v2d *>(row_t + 32) = (v2d){xtb[0], xtb[1]};
                *reinterpret_cast<v2d *>(row_t + 34) = (v2d){xtb[2], xtb[3]};
                *reinterpret_cast<v2d *>(row_b) = (v2d){xbt[0], xbt[1]};
                *reinterpret_cast<v2d *>(row_b + 2) = (v2d){xbt[2], xbt[3]};
                *reinterpret_cast<v2d *>(row_b + 32) = (v2d){xbb[0], xbb[1]};
                *reinterpret_cast<v2d *>(row_b + 34) = (v2d){xbb[2], xbb[3]};
            }
            EJ_MARK(2)
            __syncthreads();
            EJ_MARK(3)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                xtt[m] = X[rt + ct[m]], xtb[m] = X[rt + cb[m]];
                xbt[m] = X[rb + ct[m]], xbb[m] = X[rb + cb[m]];
            }
            {   // V's columns move with their indices: seat s takes the column of seat ej_src(s); neighbours are one lane away
                const double lt = __shfl_up(vtt[3], 1), lb = __shfl_up(vbt[3], 1);
                const double rt_ = __shfl_down(vtb[0], 1), rb_ = __shfl_down(vbb[0], 1);
                const double t0 = vtt[0], t3 = vtt[3], b0 = vbt[0], b3 = vbt[3], tb0 = vtb[0], bb0 = vbb[0];
#pragma unroll
                for (int m = 3; m >= 1; --m) vtt[m] = vtt[m - 1], vbt[m] = vbt[m - 1];
#pragma unroll
                for (int m = 0; m < 3; ++m) vtb[m] = vtb[m + 1], vbb[m] = vbb[m + 1];
                vtt[0] = lt, vbt[0] = lb, vtb[3] = rt_, vbb[3] = rb_;
                if (cq == 0) vtt[0] = t0, vbt[0] = b0, vtt[1] = tb0, vbt[1] = bb0;   // seat 0 keeps its index, seat 1 takes seat 32's
                if (cq == 7) vtb[3] = t3, vbb[3] = b3;                               // seat 63 takes seat 31's
            }
            EJ_MARK(4)
        }
        if (made_here) atomicAdd(n_rot, made_here);             // rotations in the low half, the loud ones among them in the high half
        __syncthreads();
        const int made = *n_rot;
        __syncthreads();
        if (t == 0) *n_rot = 0;                                 // (the next count is 126 barriers away)
        // Done when a sweep rotates nothing -- or only quietly: with every entry at most eps = 2^-32 of its diagonal pair when
        // its turn came and every sine at most 2^-32, what the sweep's other rotations put back into a cleared entry is at
        // most n eps^2 = 2^-58, below the 2^-53 at which a pair is left alone; the sweep that would find that out (a ninth
        // of the kernel's time) is not run.
        if ((made >> 16) == 0) break;
        if (sweep == max_sweeps - 1 && t == 0) atomicOr(status, 1);   // still rotating after the last sweep: the engine fails the call
    }
#ifdef EJ_PROF
    if (blockIdx.x == 0 && (t == 0 || t == 255 || t == 9))
        printf("eigen_kernel t=%d sweeps %d; cycles: rotation %lld, barrier %lld, rotate+store %lld, barrier %lld, load+shuffle %lld, loop %lld\n", t,
               pf_sweeps, pf[0], pf[1], pf[2], pf[3], pf[4], pf[5]);
    pf_t = clock64();
#endif
    // every index is back in its first seat: seat s = matrix index s
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        Vs[a * EJ_LD + 4 * cq + m] = vtt[m], Vs[a * EJ_LD + 32 + 4 * cq + m] = vtb[m];
        Vs[(32 + a) * EJ_LD + 4 * cq + m] = vbt[m], Vs[(32 + a) * EJ_LD + 32 + 4 * cq + m] = vbb[m];
    }
    __syncthreads();

    // ---- Newton-Schulz: E = (I - V^T V) / 2 into X, then V += V E ----
    const int row = t >> 2, l0 = (t & 3) * 16;                  // 16 entries of one row of a 64 x 64 product
    {
        double g[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) g[x] = 0.0;
        if (row < n)
            for (int i = 0; i < n; ++i) {
                const double vk = Vs[i * EJ_LD + row];
#pragma unroll
                for (int x = 0; x < 16; ++x) g[x] += vk * Vs[i * EJ_LD + l0 + x];
            }
#pragma unroll
        for (int x = 0; x < 16; ++x)
            X[row * EJ_LD + l0 + x] = (row < n && l0 + x < n) ? 0.5 * ((row == l0 + x ? 1.0 : 0.0) - g[x]) : 0.0;
    }
    __syncthreads();
    {
        double w[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) w[x] = 0.0;
        if (row < n)
            for (int kk = 0; kk < n; ++kk) {
                const double vik = Vs[row * EJ_LD + kk];
#pragma unroll
                for (int x = 0; x < 16; ++x) w[x] += vik * X[kk * EJ_LD + l0 + x];
            }
        __syncthreads();                    // all reads of V and E done before anybody updates V or overwrites E
        if (row < n)
#pragma unroll
            for (int x = 0; x < 16; ++x)
                if (l0 + x < n) Vs[row * EJ_LD + l0 + x] += w[x];
    }
    // ---- eigenvalues: lambda_l = sum_i V[i][l] (S V)[i][l] ----
    for (int idx = t; idx < NS * NS; idx += 256) X[(idx >> 6) * EJ_LD + (idx & 63)] = Sm[idx];
    __syncthreads();
    {
        double w[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) w[x] = 0.0;
        if (row < n)
            for (int kk = 0; kk < n; ++kk) {
                const double sik = X[row * EJ_LD + kk];
#pragma unroll
                for (int x = 0; x < 16; ++x) w[x] += sik * Vs[kk * EJ_LD + l0 + x];
            }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 16; ++x) X[row * EJ_LD + l0 + x] = row < n ? w[x] * Vs[row * EJ_LD + l0 + x] : 0.0;
    }
    __syncthreads();
    if (t < NS) {
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum += X[i * EJ_LD + t];
        lam[t] = t < n ? sum : 0.0;
    }
    __syncthreads();

    // ---- pack: U^T[k][i] = V[i][k] / sqrt(pi_i), U^-1[k][j] = V[j][k] sqrt(pi_j), lambda[k]; zero on the padding ----
    double *out = pool + (size_t)slot[blockIdx.x] * EIG_STRIDE;
    for (int idx = t; idx < NS * NS; idx += 256) {
        const int kk = idx >> 6, i = idx & 63;
        const bool real = kk < n && i < n;
        const double v = real ? Vs[i * EJ_LD + kk] : 0.0, sq = real ? sqrt_pi[i] : 1.0;
        out[EIG_UT + idx] = real ? v / sq : 0.0;
        out[EIG_UINV + idx] = real ? v * sq : 0.0;
    }
    if (t < NS) out[EIG_LAM + t] = lam[t];
#ifdef EJ_PROF
    if (blockIdx.x == 0 && t == 0) printf("eigen_kernel epilogue %lld cycles\n", clock64() - pf_t);
#endif
}

void launch_eigen(const double *S, const int32_t *slot, const double *sqrt_pi, double *pool, int M, int n_states, int *status,
                  hipStream_t stream)
{
    if (M <= 0) return;
    // more than the 64 KiB a kernel gets without asking (per device: the attribute belongs to the loaded code object);
    // if this fails the launch does too, and the engine reports that
    // 24 sweeps are far more than a symmetric matrix needs (9-10 at n = 61); CRAB_EIGEN_MAX_SWEEPS exists for the test of the failure path
    static const int max_sweeps = [] {
        const char *sv = std::getenv("CRAB_EIGEN_MAX_SWEEPS");
        const int v = sv ? std::atoi(sv) : 0;
        return v >= 1 && v <= 24 ? v : 24;
    }();
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(eigen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EJ_LDS_BYTES);
    hipLaunchKernelGGL(eigen_kernel, dim3(M), dim3(256), EJ_LDS_BYTES, stream, S, slot, sqrt_pi, pool, n_states, max_sweeps, status);
}

// =============================================================================================
// kernel 2: pruning
// =============================================================================================
// Register plan.  The conditional-likelihood tiles are the only large state of the walk, they live for the whole
// program, and every op updates them in place.  Left to the register allocator they are re-homed at every merge
// point of the op-kind branches (hundreds of v_mov_b64 and scratch spills per op).  So the compiler is confined to
// v[0, CC) (amdgpu_num_vgpr) and the tiles and the A-operand ring sit in fixed registers above it, touched only by
// the asm statements below, which name them through "n" (immediate) operands:
//     v[AV, AV+32)        ring of eight 16-byte A-operand reads
//     v[T0, T0 + 32 NT)   tile 0: element (nt, idx) in v[T0 + 2 (16 nt + idx) : +1]
//     v[T1, T1 + 32 NT)   tile 1
// Element idx of pattern group nt holds state 4 idx + (lane >> 4) of pattern lane & 15 -- the C/D and the B lane map
// of v_mfma_f64_4x4x4_4b_f64.  Since these registers are invisible to the compiler, every asm that loads into them
// is followed by an explicit s_waitcnt before the first use.
// With NT = 1 (256 registers, two waves per SIMD) there is room for more tiles: G receives the columns of the next
// leaf edge while products run (the gather's L2 round trip hides under MFMA work), and S0, S1 hold stack slots 0 and 1
// of the saved sibling products -- the host numbers the slots by how often the tree writes them, so these are the two
// busiest frame depths; on trees of Sethi-Ullman depth <= 2 no conditional likelihood ever leaves the register
// file.  The other slots (and every slot with NT = 2) go to the global-memory stack.
constexpr int N_XCD = 8;  // the launch grid is a multiple of this
#ifndef CRAB_PRIO_CORE   // issue priorities of a wave inside a product and outside one (experiments: -DCRAB_PRIO_CORE=.. -DCRAB_PRIO_OTHER=..)
#define CRAB_PRIO_CORE 0
#define CRAB_PRIO_OTHER 1
#endif
// -DCRAB_NO_PATH_OPS compiles the save / load ops of partial updates and tree parts out of the kernel (same-box A/B of
// what their presence costs a plain walk; such a build must run with CRAB_NO_PATH_UPDATES=1 CRAB_NO_TREE_PARTS=1)
#ifdef CRAB_NO_PATH_OPS
constexpr bool PATH_OPS = false;
#else
constexpr bool PATH_OPS = true;
#endif

template <int NT>
struct Plan {
#ifdef CRAB_PROF
    static constexpr int CC = 96;  // the marks push the compiler a few registers past 64: one register-resident stack slot less
#else
    static constexpr int CC = NT == 1 ? 64 : 96;  // registers the compiler may allocate
#endif
    static constexpr int AV = CC;
    static constexpr int T0 = AV + 32;
    static constexpr int T1 = T0 + 32 * NT;
    static constexpr bool HAS_PRE = NT == 1;      // a prefetch tile G for the next leaf edge's columns
    static constexpr int G = T1 + 32 * NT;
    static constexpr int S0 = G + (HAS_PRE ? 32 * NT : 0);
    static constexpr int NSTK = NT == 1 ? (256 - S0) / 32 : 0;  // stack slots held in registers
    static constexpr int TOP = S0 + 32 * NT * NSTK;
    static_assert(TOP <= 256, "register plan exceeds the two-waves-per-SIMD budget");
};


// ---- one-instruction building blocks (register numbers are template arguments) ----
template <int D, int A, int B, bool FIRST>
__device__ __forceinline__ void r_mfma()
{
#ifdef CRAB_ABLATE_MFMA
    return;
#endif
    if constexpr (FIRST)
        asm volatile("v_mfma_f64_4x4x4_4b_f64 v[%0:%1], v[%2:%3], v[%4:%5], 0" ::"n"(D), "n"(D + 1), "n"(A), "n"(A + 1), "n"(B), "n"(B + 1));
    else
        asm volatile("v_mfma_f64_4x4x4_4b_f64 v[%0:%1], v[%2:%3], v[%4:%5], v[%0:%1]" ::"n"(D), "n"(D + 1), "n"(A), "n"(A + 1), "n"(B), "n"(B + 1));
}
template <int D, int OFF>
__device__ __forceinline__ void r_lds_read128(unsigned addr)
{
    asm volatile("ds_read_b128 v[%0:%1], %2 offset:%3" ::"n"(D), "n"(D + 3), "v"(addr), "n"(OFF) : "memory");
}
template <int X, int Y>
__device__ __forceinline__ void r_mul()
{
    asm volatile("v_mul_f64 v[%0:%1], v[%0:%1], v[%2:%3]" ::"n"(X), "n"(X + 1), "n"(Y), "n"(Y + 1));
}
template <int D, int S>
__device__ __forceinline__ void r_copy()
{
    asm volatile("v_mov_b64 v[%0:%1], v[%2:%3]" ::"n"(D), "n"(D + 1), "n"(S), "n"(S + 1));
}
template <int X>
__device__ __forceinline__ void r_scale(double sc)
{
    asm volatile("v_mul_f64 v[%0:%1], v[%0:%1], %2" ::"n"(X), "n"(X + 1), "v"(sc));
}
template <int X>
__device__ __forceinline__ void r_set(double v)
{
    asm volatile("v_mov_b64 v[%0:%1], %2" ::"n"(X), "n"(X + 1), "v"(v));
}
template <int X>
__device__ __forceinline__ void r_fma_into(double &f, double m)
{
    asm volatile("v_fma_f64 %0, %1, v[%2:%3], %0" : "+v"(f) : "v"(m), "n"(X), "n"(X + 1));
}
template <int D, int OFF>
__device__ __forceinline__ void r_gload128(const void *sbase, unsigned voff)
{
    asm volatile("global_load_dwordx4 v[%0:%1], %2, %3 offset:%4" ::"n"(D), "n"(D + 3), "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int D, int OFF>
__device__ __forceinline__ void r_gload64(const void *sbase, unsigned voff)
{
    asm volatile("global_load_dwordx2 v[%0:%1], %2, %3 offset:%4" ::"n"(D), "n"(D + 1), "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int S, int OFF>
__device__ __forceinline__ void r_gstore64(void *sbase, unsigned voff)
{
    asm volatile("global_store_dwordx2 %2, v[%0:%1], %3 offset:%4" ::"n"(S), "n"(S + 1), "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int S, int OFF>
__device__ __forceinline__ void r_gstore128(void *sbase, unsigned voff)
{
    asm volatile("global_store_dwordx4 %2, v[%0:%1], %3 offset:%4" ::"n"(S), "n"(S + 3), "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// A wave-uniform pointer, provably so: the "s" operands of the loads and stores above must come out in scalar registers
// whatever the divergence analysis makes of the loops around them (folds away when the value is in scalar registers already).
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
// The hazard recognizer does not look inside inline asm: a vector-memory instruction that reads a scalar register written by a
// VALU instruction (v_readlane, v_readfirstlane: how a spilled or "proven" wave-uniform base gets back into scalar registers)
// needs five wait states in between, which the compiler inserts for its own instructions only.  One s_nop 4 in front of every
// asm sequence that takes a scalar base.  (A build whose bases all came through v_readfirstlane faulted on the GPU.)
__device__ __forceinline__ void r_sgpr_hazard() { asm volatile("s_nop 4" ::: "memory"); }
__device__ __forceinline__ void r_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- tile-wide operations ----
template <int X, int Y, int... I>
__device__ __forceinline__ void t_mul_seq(std::integer_sequence<int, I...>)
{
    (r_mul<X + 2 * I, Y + 2 * I>(), ...);
}
// X *= Y, element-wise, NT groups
template <int NT, int X, int Y>
__device__ __forceinline__ void t_mul()
{
    t_mul_seq<X, Y>(std::make_integer_sequence<int, 16 * NT>{});
}
template <int D, int S, int... I>
__device__ __forceinline__ void t_copy_seq(std::integer_sequence<int, I...>)
{
    (r_copy<D + 2 * I, S + 2 * I>(), ...);
}
// D = S, element-wise, NT groups
template <int NT, int D, int S>
__device__ __forceinline__ void t_copy()
{
    t_copy_seq<D, S>(std::make_integer_sequence<int, 16 * NT>{});
}
template <int X, int... I>
__device__ __forceinline__ void t_scale_seq(double sc, std::integer_sequence<int, I...>)
{
    (r_scale<X + 2 * I>(sc), ...);
}
// childless non-leaf node: ones (or zeros) on the real states, zeros on the padding
template <int X, int... I>
__device__ __forceinline__ void t_const_seq(int ones, int g, int n_states, std::integer_sequence<int, I...>)
{
    (r_set<X + 2 * I>((ones && 4 * I + g < n_states) ? 1.0 : 0.0), ...);
}
template <int X, int... I>
__device__ __forceinline__ void t_dot_seq(double &f, const double (&m)[16], std::integer_sequence<int, I...>)
{
    (r_fma_into<X + 2 * I>(f, m[I]), ...);
}
// 16-byte column pieces of one pattern group: eight loads of 16 bytes, 64 bytes apart
template <int D, int... Q>
__device__ __forceinline__ void t_gather_seq(const void *sbase, unsigned voff, std::integer_sequence<int, Q...>)
{
    (r_gload128<D + 4 * Q, 64 * Q>(sbase, voff), ...);
}
#ifdef CRAB_ABLATE_COALESCE
template <int D, int... Q>
__device__ __forceinline__ void t_gather_block_seq(const void *sbase, unsigned voff, std::integer_sequence<int, Q...>)
{
    (r_gload128<D + 4 * Q, 1024 * (Q & 3)>(sbase, voff + (Q >> 2) * 4096), ...);
}
#endif
// stack slot <-> tile.  A saved tile is laid out like a block of 16 gather-image columns (pattern c at c * 512 bytes,
// the lane's 16-byte piece q at 64 q + 16 g; voff = (lane & 15) * 512 + (lane >> 4) * 16): eight 16-byte accesses per
// lane.  (In this layout a pop can also ride inside the product that precedes it, through Ride::img_b; measured
// neutral -- the other wave of the SIMD already covers the load -- and left out.)
template <int S, int... Q>
__device__ __forceinline__ void t_scatter_seq(void *sbase, unsigned voff, std::integer_sequence<int, Q...>)
{
    (r_gstore128<S + 4 * Q, 64 * Q>(sbase, voff), ...);
}
template <int NT, int D>
__device__ __forceinline__ void t_load(const void *sbase, unsigned voff)
{
    using S8 = std::make_integer_sequence<int, 8>;
    r_sgpr_hazard();
    t_gather_seq<D>(sbase, voff, S8{});
    if constexpr (NT == 2) t_gather_seq<D + 32>(sbase, voff + 8192, S8{});
}
template <int NT, int S>
__device__ __forceinline__ void t_store(void *sbase, unsigned voff)
{
    using S8 = std::make_integer_sequence<int, 8>;
    r_sgpr_hazard();
    t_scatter_seq<S>(sbase, voff, S8{});
    if constexpr (NT == 2) t_scatter_seq<S + 32>(sbase, voff + 8192, S8{});
}

// Largest high word over the 16 elements of group nt (conditional likelihoods are non-negative, so the largest
// value is the one with the largest high word), then over the four lanes c, c+16, c+32, c+48 that share a pattern.
template <int X>
__device__ __forceinline__ int t_max_hi()
{
    int hi;
    asm volatile("v_max3_i32 %0, v[%1], v[%2], v[%3]" : "=v"(hi) : "n"(X + 1), "n"(X + 3), "n"(X + 5));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 7), "n"(X + 9));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 11), "n"(X + 13));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 15), "n"(X + 17));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 19), "n"(X + 21));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 23), "n"(X + 25));
    asm volatile("v_max3_i32 %0, %0, v[%1], v[%2]" : "+v"(hi) : "n"(X + 27), "n"(X + 29));
    asm volatile("v_max_i32 %0, %0, v[%1]" : "+v"(hi) : "n"(X + 31));
    auto r16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);  // row swaps: no LDS traffic
    hi = (int)r16[0] > (int)r16[1] ? (int)r16[0] : (int)r16[1];
    auto r32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return (int)r32[0] > (int)r32[1] ? (int)r32[0] : (int)r32[1];
}
// Normalise tile X so that each pattern's maximum over the 64 states lies in [1, 2): multiply by an exact power of
// two and book the exponent.
template <int NT, int X>
__device__ __forceinline__ void t_normalise(int (&e)[NT])
{
    static_assert(NT <= 2, "");
    int hi[NT];
    hi[0] = t_max_hi<X>();
    if constexpr (NT == 2) hi[1] = t_max_hi<X + 32>();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int ex = ((hi[nt] >> 20) & 0x7ff) - 1023;  // biased exponent of the maximum
        const bool ok = hi[nt] >= 0x00100000 && hi[nt] < 0x7ff00000;  // positive, normal, finite
        ex = ok ? (ex > 1022 ? 1022 : ex) : 0;
        const double sc = __hiloint2double((1023 - ex) << 20, 0);
        if (nt == 0)
            t_scale_seq<X>(sc, std::make_integer_sequence<int, 16>{});
        else
            t_scale_seq<X + 32>(sc, std::make_integer_sequence<int, 16>{});
        e[nt] += ex;
    }
}

// Issue the LDS-DMA of one 32 KiB operand image: each of the 4 waves moves 8 x 1 KiB.
__device__ __forceinline__ void dma_matrix(const double *Pm, char *lds_buf, int wave, int lane)
{
    const char *gsrc = (const char *)Pm + wave * 8192 + lane * 16;
    char *ldst = lds_buf + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc + i * 1024), LDS_PTR(ldst + i * 1024), 16, 0, 0);
}

// Y = P * X for NT pattern groups: 256*NT MFMAs fed by 128 ds_read_b128, P's operand image in LDS at lds_addr.
// Group GI = (sp, ic) covers k-tiles 2sp, 2sp+1 and state groups 4ic..4ic+3.  Read q of group GI+1 is issued just
// before the MFMAs that consume read q of group GI, so every read has 8*NT MFMAs (>= 128 cycles) to land; between
// the two MFMAs into one accumulator lie 4*NT independent ones.  Measured alone: 74-76 TFLOP/s (scripts/micro).
template <int NT, int X, int Y, int GI, int Q>
__device__ __forceinline__ void gemm_first_half(unsigned lds_addr)
{
    constexpr int R_AV = Plan<NT>::AV;
    constexpr int sp = GI >> 2, ic = GI & 3, b = GI & 1;
    if constexpr (GI + 1 < 32) {
        constexpr int sp1 = (GI + 1) >> 2, ic1 = (GI + 1) & 3;
        r_lds_read128<R_AV + ((b ^ 1) * 4 + Q) * 4, ((ic1 * 4 + Q) * 8 + sp1) * 256>(lds_addr);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 - Q) : "memory");
    }
    constexpr int A = R_AV + (b * 4 + Q) * 4;
#ifdef CRAB_SKIP15_EMU   // timing emulation only (scripts/micro/gemm_tile_rate.hip): what a 15 x 15 block product would issue
    if constexpr (ic * 4 + Q != 15)
#endif
    {
        r_mfma<Y + 2 * (ic * 4 + Q), A, X + 2 * (2 * sp), sp == 0>();
        if constexpr (NT == 2) r_mfma<Y + 32 + 2 * (ic * 4 + Q), A, X + 32 + 2 * (2 * sp), sp == 0>();
    }
    if constexpr (Q + 1 < 4) gemm_first_half<NT, X, Y, GI, Q + 1>(lds_addr);
}
template <int NT, int X, int Y, int GI, int Q>
__device__ __forceinline__ void gemm_second_half()
{
    constexpr int R_AV = Plan<NT>::AV;
    constexpr int sp = GI >> 2, ic = GI & 3, b = GI & 1;
    constexpr int A = R_AV + (b * 4 + Q) * 4 + 2;
#ifdef CRAB_SKIP15_EMU
    if constexpr (ic * 4 + Q != 15 && 2 * sp + 1 != 15)
#endif
    {
        r_mfma<Y + 2 * (ic * 4 + Q), A, X + 2 * (2 * sp + 1), false>();
        if constexpr (NT == 2) r_mfma<Y + 32 + 2 * (ic * 4 + Q), A, X + 32 + 2 * (2 * sp + 1), false>();
    }
    if constexpr (Q + 1 < 4) gemm_second_half<NT, X, Y, GI, Q + 1>();
}
// Vector-memory traffic that rides inside a product, one instruction every second group (= 16*NT MFMAs), so that the
// wave never queues a burst of loads in front of its MFMAs: the eight LDS-DMA pieces of the next product's operand
// image in groups 0..14, then (kernels with a prefetch tile) the eight column loads of the next leaf edge in 16..30.
template <int G, int XB>
struct Ride {
    const char *dma_src;   // this lane's first 16 bytes of the image to stage, or nullptr
    char *dma_dst;         // this wave's 8 KiB of the LDS buffer it goes to
    const char *img;       // gather image of the leaf edge to prefetch into G (wave-uniform), or nullptr
    unsigned voff;         // this lane's byte offset in it
    const char *img_b;     // gather image of a second leaf edge, prefetched into the B operand tile XB as its
    unsigned voff_b;       //   k-tiles die (piece q overwrites k-tiles 2q, 2q+1 after group 4q+3 has issued), or nullptr
    template <int GI>
    __device__ __forceinline__ void at() const
    {
        if constexpr (GI % 2 == 1 && GI < 16) {
            if (dma_src) {
                constexpr int i = GI / 2;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(dma_src + i * 1024), LDS_PTR(dma_dst + i * 1024), 16, 0, 0);
                asm volatile("" ::: "memory");
            }
        } else if constexpr (GI % 2 == 1 && G >= 0) {
            if (img) {
                constexpr int q = (GI - 17) / 2;
#ifdef CRAB_ABLATE_COALESCE
                r_gload128<(G >= 0 ? G : 0) + 4 * q, 1024 * (q & 3)>(img, voff + (q >> 2) * 4096);
#else
                r_gload128<(G >= 0 ? G : 0) + 4 * q, 64 * q>(img, voff);
#endif
            }
        } else if constexpr (GI % 4 == 0 && GI >= 4 && G >= 0) {
            if (img_b) {
                constexpr int q = GI / 4 - 1;
#ifdef CRAB_ABLATE_COALESCE
                r_gload128<XB + 4 * q, 1024 * (q & 3)>(img_b, voff_b + (q >> 2) * 4096);
#else
                r_gload128<XB + 4 * q, 64 * q>(img_b, voff_b);
#endif
            }
        }
    }
    __device__ __forceinline__ void after() const
    {
        if constexpr (G >= 0)
#ifdef CRAB_ABLATE_COALESCE
            if (img_b) r_gload128<XB + 28, 1024 * 3>(img_b, voff_b + 4096);
#else
            if (img_b) r_gload128<XB + 28, 64 * 7>(img_b, voff_b);
#endif
    }
};
template <int NT, int X, int Y, int GI, typename R>
__device__ __forceinline__ void gemm_group(unsigned lds_addr, const R &ride)
{
    ride.template at<GI>();
    gemm_first_half<NT, X, Y, GI, 0>(lds_addr);
    gemm_second_half<NT, X, Y, GI, 0>();
    if constexpr (GI + 1 < 32) gemm_group<NT, X, Y, GI + 1>(lds_addr, ride);
}
template <int NT, int X, int Y, typename R>
__device__ __forceinline__ void gemm_tile(unsigned lds_addr, const R &ride)
{
    constexpr int R_AV = Plan<NT>::AV;
    r_lds_read128<R_AV + 0, 0 * 2048>(lds_addr);
    r_lds_read128<R_AV + 4, 1 * 2048>(lds_addr);
    r_lds_read128<R_AV + 8, 2 * 2048>(lds_addr);
    r_lds_read128<R_AV + 12, 3 * 2048>(lds_addr);
    gemm_group<NT, X, Y, 0>(lds_addr, ride);
    ride.after();
    // the hardware does not interlock MFMA results against the VALU / memory instructions that read them next
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

// Build with -DCRAB_CLOCK to measure the shader clock the chip holds under this kernel: one s_memtime / s_memrealtime
// pair around each wave's whole life (nothing inside the walk), printed by launch_prune.  Diagnostic builds only.
#ifdef CRAB_CLOCK
__device__ unsigned long long *g_clock_buf;
#endif
// Build with -DCRAB_PROF to get a per-phase cycle breakdown of the tree walk (s_memtime, summed per wave, printed by
// launch_prune to stderr).  Diagnostic builds only: scripts/gpu_prof_phases.sh.  The marks must cost the compiler next to
// nothing: the twelve 64-bit counters of a wave are the 24 low lanes of ONE vector register (v_readlane / scalar add /
// v_writelane) plus one 64-bit scalar for the last mark.  Even so the allocator ends up at v69 where the shipped build ends at
// v63 (amdgpu_num_vgpr is a request: asking for 40 or 48 changes nothing), so this build gives the compiler 96 registers and
// the walk one register-resident stack slot instead of two.  (Round 2 kept the counters in 26 scalar pairs; the compiler overran
// even the 96 into the A-operand ring, and the first product's LDS reads overwrote spilled pointers: an illegal access in the
// first launch.  scripts/check_reserved_vgprs.py -DCRAB_PROF sees exactly that, and runs as a test now.)
#ifdef CRAB_PROF
constexpr int PROF_N = 12;
template <int I>
__device__ __forceinline__ void prof_add(unsigned &acc, unsigned long long d)
{
    const unsigned long long sum = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)acc, 2 * I + 1) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)acc, 2 * I)) + d;
    // (readfirstlane: the "s" operands must be provably wave-uniform for the compiler to put them in scalar registers)
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sum), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sum >> 32));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(lo), "n"(2 * I));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(hi), "n"(2 * I + 1));
}
#define PROF_T(i)                                                   \
    {                                                               \
        const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
        prof_add<i>(w.prof_acc, _t - w.prof_last);                  \
        w.prof_last = _t;                                           \
    }
#else
#define PROF_T(i)
#endif

// State of one wave's walk that is not a register tile.
template <int NT>
struct Walk {
    const PruneArgs &a;
    const DevOp *ops;          // this class's program
    DevOp batch;               // 64 ops, one per lane, decoded with v_readlane
    int ip;                    // next op
    int buf;                   // LDS buffer holding the operand image of the next product
    unsigned code_off[NT];     // code queue: byte offset of the column inside a gather image, per pattern group, for
    unsigned code_off2[NT];    //   the next leaf edge whose columns are not requested yet, and the one after it
    const uint8_t *codes;      // the code table this program reads (wave-uniform: per-lane state is kept to 32-bit offsets, the
    unsigned pat;              //   compiler's registers are few); this lane's pattern in group 0 = its column there and in idx32
    size_t table_off;          // this wave's 16 columns inside a clade table: (tile * 4 + wave) * 8192 bytes
    const char *gather0;       // gather image of matrix 0
    char *stack;               // this wave's stack area (uniform) ...
    int32_t *stack_e;          // ... and its exponent area (uniform; this lane's element is [lane])
    int estk[3][NT];           // exponents of the register-resident stack slots (first NSTK used)
    unsigned stack_voff;       // this lane's byte offset inside a saved tile: (lane & 15) * 512 + (lane >> 4) * 16
    unsigned lds_a;            // LDS byte address of this lane's A operand inside a tile pair
    char *smem;
    int wave, lane;
    int n_ops;                 // length of this program
    char *save0;               // this wave's tile of save slot 0 (uniform), or nullptr; slot s is s * save_stride further
    size_t save_stride;
    bool more;                 // the work item has another tile after this one (F_NEXT_TILE, ROW_NEXT_TILE)
#ifdef CRAB_PROF
    unsigned prof_acc;              // lanes 2i, 2i + 1: this wave's counter i
    unsigned long long prof_last;
#endif
};

// A header or item record through the constant address space (sixteen scalar dwords).
__device__ __forceinline__ ProgHdr load_hdr(const ProgHdr *p)
{
    ProgHdr h;
    const auto *src = CONST_PTR(int32_t, p);
    int32_t *dst = reinterpret_cast<int32_t *>(&h);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(ProgHdr) / 4); ++i) dst[i] = src[i];
    return h;
}
// An item record from the LDS mailbox (put there by post_next_number's DMA while the previous item ran).
typedef int int4_lds __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ProgHdr read_hdr_lds(unsigned addr)
{
    int4_lds q0, q1, q2, q3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                 : "v"(addr)
                 : "memory");
    ProgHdr h;
    int32_t *dst = reinterpret_cast<int32_t *>(&h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[i] = __builtin_amdgcn_readfirstlane(q0[i]);
        dst[4 + i] = __builtin_amdgcn_readfirstlane(q1[i]);
        dst[8 + i] = __builtin_amdgcn_readfirstlane(q2[i]);
        dst[12 + i] = __builtin_amdgcn_readfirstlane(q3[i]);
    }
    return h;
}
// Byte offset, inside a gather image, of this lane's share of the column its pattern shows in alignment row `row`.
template <int NT>
__device__ __forceinline__ void load_code_offsets(const Walk<NT> &w, int row, unsigned (&off)[NT])
{
    unsigned code[NT];
    size_t adv = 0;
    if (row & ROW_NEXT_TILE) {   // the same row, one tile on (wave-uniform)
        if (!w.more) return;
        row &= ~ROW_NEXT_TILE;
        adv = 64 * NT;
    }
    if (w.a.n_code_rows > 0 && row >= w.a.n_code_rows) {   // a clade's table: the column of this pattern's sub-pattern (wave-uniform branch)
        const uint32_t *r = w.a.idx32 + (size_t)(row - w.a.n_code_rows) * w.a.n_pat_pad + adv;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#ifdef CRAB_ABLATE_CODES
            code[nt] = (unsigned)(row + (w.lane & 15) + nt) % 61u;
            (void)r;
#else
            code[nt] = r[w.pat + nt * 16];
#endif
        }
    } else {
        const uint8_t *r = w.codes + (size_t)row * w.a.n_pat_pad + adv;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#ifdef CRAB_ABLATE_CODES   // timing experiment only (wrong results): no load, so no wait for one either
            code[nt] = (unsigned)(row + (w.lane & 15) + nt) % 61u;
            (void)r;
#else
            code[nt] = r[w.pat + nt * 16];
#endif
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#ifdef CRAB_ABLATE_COALESCE
        off[nt] = ((unsigned)__builtin_amdgcn_readfirstlane((int)code[nt]) & ~15u) * 512u + (unsigned)w.lane * 16u;
#else
        off[nt] = code[nt] * 512u + (unsigned)(w.lane >> 4) * 16u;
#endif
    }
}
// The code queue holds the offsets of the next two leaf edges whose columns have not been requested yet
// (code_off, code_off2).  One request took the head: shift, and fetch the codes of alignment row `row` behind it.
template <int NT>
__device__ __forceinline__ void refill_codes1(Walk<NT> &w, int row)
{
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w.code_off[nt] = w.code_off2[nt];
    if (row >= 0) load_code_offsets<NT>(w, row, w.code_off2);
}
// Two requests took both entries.
template <int NT>
__device__ __forceinline__ void refill_codes2(Walk<NT> &w, int row_a, int row_b)
{
    if (row_a >= 0) load_code_offsets<NT>(w, row_a, w.code_off);
    if (row_b >= 0) load_code_offsets<NT>(w, row_b, w.code_off2);
}
// Requests P[i][code] for the 16 states of this lane into tile D, from the gather image of matrix `mat` (L2-resident,
// shared by every workgroup walking this class), with the codes at the head of the queue.
template <int NT, int D>
__device__ __forceinline__ void issue_columns(Walk<NT> &w, int mat)
{
    // (readfirstlane: the "s" operand of the loads must be provably wave-uniform)
    const char *img = uniform_ptr(w.gather0 + (size_t)__builtin_amdgcn_readfirstlane(mat) * (P_ELEMS * 8));
    using S8 = std::make_integer_sequence<int, 8>;
    r_sgpr_hazard();
#ifdef CRAB_ABLATE_GATHER
    (void)img;
#else
#ifdef CRAB_ABLATE_COALESCE   // the same 8 KiB per tile as one contiguous block: what the scattered 64-byte pieces cost
    t_gather_block_seq<D>(img, w.code_off[0], S8{});
#else
    t_gather_seq<D>(img, w.code_off[0], S8{});
    if constexpr (NT == 2) t_gather_seq<D + 32>(img, w.code_off[1], S8{});
#endif
#endif
}
// Kernels without a prefetch tile: columns now, then the codes of the next leaf edge (alignment row `row_next`).
template <int NT, int D>
__device__ __forceinline__ void issue_leaf(Walk<NT> &w, int mat, int row_next)
{
    issue_columns<NT, D>(w, mat);
    if (row_next >= 0) load_code_offsets<NT>(w, row_next, w.code_off);
}

// Saves the running product in tile X in stack slot `slot`: a register tile for the first NSTK slots, global memory
// beyond.  The exponents of the register slots live in w.estk (compiler registers).
template <int NT, int X>
__device__ __forceinline__ void push(Walk<NT> &w, int slot, const int (&eX)[NT])
{
    using P = Plan<NT>;
    constexpr size_t slot_bytes = (size_t)NT * 16 * 64 * 8;
    if constexpr (P::NSTK >= 1)
        if (slot == 0) {
            t_copy<NT, P::S0, X>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w.estk[0][nt] = eX[nt];
            return;
        }
    if constexpr (P::NSTK >= 2)
        if (slot == 1) {
            t_copy<NT, P::S0 + 32 * NT, X>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w.estk[1][nt] = eX[nt];
            return;
        }
    if constexpr (P::NSTK >= 3)
        if (slot == 2) {
            t_copy<NT, P::S0 + 64 * NT, X>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w.estk[2][nt] = eX[nt];
            return;
        }
    int32_t *se = w.stack_e + (size_t)slot * (NT * 64) + w.lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) se[nt * 64] = eX[nt];
    t_store<NT, X>(uniform_ptr(w.stack + (size_t)slot * slot_bytes), w.stack_voff);
}
// Y *= stack[slot]; the dead tile L is the landing zone when the slot is in global memory.
template <int NT, int Y, int L>
__device__ __forceinline__ void pop_mul(Walk<NT> &w, int slot, int (&eY)[NT])
{
    using P = Plan<NT>;
    constexpr size_t slot_bytes = (size_t)NT * 16 * 64 * 8;
    if constexpr (P::NSTK >= 1)
        if (slot == 0) {
            t_mul<NT, Y, P::S0>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) eY[nt] += w.estk[0][nt];
            return;
        }
    if constexpr (P::NSTK >= 2)
        if (slot == 1) {
            t_mul<NT, Y, P::S0 + 32 * NT>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) eY[nt] += w.estk[1][nt];
            return;
        }
    if constexpr (P::NSTK >= 3)
        if (slot == 2) {
            t_mul<NT, Y, P::S0 + 64 * NT>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) eY[nt] += w.estk[2][nt];
            return;
        }
    const int32_t *se = w.stack_e + (size_t)slot * (NT * 64) + w.lane;
    int es[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) es[nt] = se[nt * 64];
    t_load<NT, L>(uniform_ptr(w.stack + (size_t)slot * slot_bytes), w.stack_voff);
    r_wait_vm0();
    t_mul<NT, Y, L>();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) eY[nt] += es[nt];
}

// Save area of the base vector's walk (crab_kernels.h): tile X and its exponents into / out of save slot `slot`.
template <int NT, int X>
__device__ __forceinline__ void save_tile(Walk<NT> &w, int slot, const int (&eX)[NT])
{
    static_assert(NT == 1, "path programs are built for the 64-pattern kernel only");
    char *dst = uniform_ptr(w.save0 + (size_t)slot * w.save_stride);
    ((int32_t *)(dst + 16 * 64 * 8))[w.lane] = eX[0];
    t_store<NT, X>(dst, w.stack_voff);
}
template <int NT, int D>
__device__ __forceinline__ void load_saved(Walk<NT> &w, int slot, int (&es)[NT])
{
    static_assert(NT == 1, "path programs are built for the 64-pattern kernel only");
    const char *src = uniform_ptr(w.save0 + (size_t)slot * w.save_stride);
    es[0] = ((const int32_t *)(src + 16 * 64 * 8))[w.lane];
    t_load<NT, D>(src, w.stack_voff);
    r_wait_vm0();
}

// Runs the program on the current tile X (exponents eX) until a product Y = P X has been executed (returns false:
// Y is the current tile now and X is free) or the program ends (returns true: X is the root's tile).
// The other tile Y is dead on entry and serves as landing zone for columns and saved products.  Every update of
// the current tile is in place and the two tiles swap roles only at a product, so no tile is ever copied.
template <int NT, int X, int Y>
__device__ __forceinline__ bool walk(Walk<NT> &w, int (&eX)[NT], int (&eY)[NT])
{
    constexpr bool PRE = Plan<NT>::HAS_PRE;
    constexpr int G = Plan<NT>::G;
    const PruneArgs &a = w.a;

    for (;;) {
        // (readfirstlane: the op counter is wave-uniform, and the compiler must know it -- it once took it for divergent and
        // decoded every field of every op in a waterfall loop)
        w.ip = __builtin_amdgcn_readfirstlane(w.ip);
        if (w.ip >= w.n_ops) break;
        PROF_T(9)
        const int il = w.ip & 63;
        if (il == 0 && w.ip > 0) w.batch = w.ops[w.ip + w.lane];
        const int w0 = __builtin_amdgcn_readlane(w.batch.w0, il);
        const int kind = w0 & 3, flags = (w0 >> 2) & ((1 << OP_FLAG_BITS) - 1), slot = w0 >> (2 + OP_FLAG_BITS);
        int sv = 0;
        if constexpr (NT == 1 && PATH_OPS) sv = __builtin_amdgcn_readlane(w.batch.sv, il);
        const int mat = __builtin_amdgcn_readlane(w.batch.mat, il);
        const int arg = __builtin_amdgcn_readlane(w.batch.arg, il);
        int pre_mat = -1, pre_row2 = -1, pre_mat_b = -1, pre_row3 = -1;
        if constexpr (PRE) {
            pre_mat = __builtin_amdgcn_readlane(w.batch.pre_mat, il);
            pre_row2 = __builtin_amdgcn_readlane(w.batch.pre_row2, il);
            pre_mat_b = __builtin_amdgcn_readlane(w.batch.pre_mat_b, il);
            pre_row3 = __builtin_amdgcn_readlane(w.batch.pre_row3, il);
        }
        ++w.ip;
        PROF_T(11)

        if (kind == OP_GEMM) {
            // this edge's operand image was requested one product ago; wait for this wave's share of it, and
            // (barrier) for everybody's: all waves have then also left the product that read the other buffer
            r_wait_vm0();
            PROF_T(1)
            __syncthreads();
            PROF_T(2)
            if constexpr (NT == 1 && PATH_OPS)
                if (flags & F_SAVE) save_tile<NT, X>(w, sv, eX);   // the operand of this edge's product
            Ride<(PRE ? G : -1), X> ride;
            // (on the last product of a cyclic program the two column requests are the head of the item's next tile and the image
            // it stages is the first product's: all three only if there is such a tile)
            const bool pre_ok = !(flags & F_NEXT_TILE) || w.more;
            const bool stage = arg >= 0 && pre_ok;
            ride.dma_src = stage ? (const char *)(a.P + (size_t)arg * P_ELEMS) + w.wave * 8192 + w.lane * 16 : nullptr;
            ride.dma_dst = w.smem + (w.buf ^ 1) * 32768 + w.wave * 8192;
            const bool pre_g = PRE && (flags & F_PRE_EARLY) && pre_ok, pre_b = PRE && (flags & F_PRE_DYING) && pre_ok;
            ride.img = uniform_ptr(pre_g ? w.gather0 + (size_t)pre_mat * (P_ELEMS * 8) : nullptr);
            ride.voff = w.code_off[0];
            ride.img_b = uniform_ptr(pre_b ? w.gather0 + (size_t)pre_mat_b * (P_ELEMS * 8) : nullptr);
            ride.voff_b = pre_g ? w.code_off2[0] : w.code_off[0];
            // Outside a product a wave is on its latency-bound critical path (gathers, multiplies, barriers) while
            // the other wave of the SIMD feeds the MFMA pipe: give that path the issue priority (measured +2 %).
#ifdef CRAB_ABLATE_GATHER
            ride.img = ride.img_b = nullptr;
#endif
#ifdef CRAB_ABLATE_DMA
            ride.dma_src = nullptr;
#endif
            __builtin_amdgcn_s_setprio(CRAB_PRIO_CORE);
            gemm_tile<NT, X, Y>(w.lds_a + w.buf * 32768, ride);
            __builtin_amdgcn_s_setprio(CRAB_PRIO_OTHER);
            if (pre_g && pre_b)
                refill_codes2<NT>(w, pre_row2, pre_row3);
            else if (pre_g || pre_b)
                refill_codes1<NT>(w, pre_row2);
            PROF_T(3)
            w.buf ^= 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) eY[nt] = eX[nt];
            if constexpr (NT == 1 && PATH_OPS) {
                if ((flags & (F_SAVE | F_POP_MUL)) == (F_SAVE | F_POP_MUL)) save_tile<NT, Y>(w, sv + 2, eY);   // the edge's message
                if (flags & F_POP_SAVED) {                     // path program: the running product the base walk saved
                    int es[NT];
                    load_saved<NT, X>(w, slot, es);
                    t_mul<NT, Y, X>();
                    eY[0] += es[0];
                } else if (flags & F_POP_MUL) {
                    pop_mul<NT, Y, X>(w, slot, eY);
                }
            } else {
                if (flags & F_POP_MUL) pop_mul<NT, Y, X>(w, slot, eY);
            }
            PROF_T(4)
            if (flags & F_RESCALE) t_normalise<NT, Y>(eY);
            PROF_T(7)
            if constexpr (NT == 1 && PATH_OPS)
                if ((flags & (F_SAVE | F_FIN)) == F_SAVE) save_tile<NT, Y>(w, sv + 1, eY);   // the parent's running product
            if (flags & F_PUSH) push<NT, Y>(w, slot, eY);
            PROF_T(8)
            return false;
        }

        if (kind == OP_LEAF) {
            if constexpr (PRE) {
                // the edge's columns were requested by an earlier op: into G, or by the preceding product into
                // what is now the landing tile
                r_wait_vm0();
                PROF_T(5)
                if (flags & F_IN_LANDING) {
                    t_mul<NT, X, Y>();
                } else if (flags & F_MUL_PROD) {
                    t_mul<NT, X, G>();
                } else {
                    t_copy<NT, X, G>();
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) eX[nt] = 0;
                }
                PROF_T(6)
            } else {
                // P[i][code] for the 16 states of this lane, from the edge's gather image (L2-resident, shared by
                // every workgroup walking this class); the codes were requested one leaf edge ago
                if (flags & F_MUL_PROD) {
                    issue_leaf<NT, Y>(w, mat, arg);
                } else {
                    issue_leaf<NT, X>(w, mat, arg);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) eX[nt] = 0;
                }
                if (arg >= 0)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT) : "memory");  // the columns have landed; the next codes may still fly
                else
                    r_wait_vm0();
                PROF_T(5)
                if (flags & F_MUL_PROD) t_mul<NT, X, Y>();
                PROF_T(6)
            }
        } else if (kind == OP_LOAD) {   // a tile another walk saved (path programs, tree parts), or (F_SAVE) saved by this one
            if constexpr (NT == 1 && PATH_OPS) {
                int es[NT];
                if ((flags & (F_SAVE | F_TABLE)) == (F_SAVE | F_TABLE)) {
                    // the clade's message for 16 sub-patterns: fold the exponents in (exact: a clade is a few leaves deep) and
                    // store the tile as 16 columns of the table the main program gathers from
                    int ex = eX[0];
                    ex = ex < -1022 ? -1022 : ex > 1023 ? 1023 : ex;
                    t_scale_seq<X>(__hiloint2double((1023 + ex) << 20, 0), std::make_integer_sequence<int, 16>{});
#ifndef CRAB_ABLATE_TABLE_STORE   // (CRAB_ABLATE_*: timing experiments with wrong results, scripts/gpu_ablate.sh)
                    t_store<NT, X>(uniform_ptr((char *)w.gather0 + (size_t)__builtin_amdgcn_readfirstlane(mat) * (P_ELEMS * 8) + w.table_off), w.stack_voff);
#endif
                } else if (flags & F_SAVE) {
                    save_tile<NT, X>(w, sv, eX);
                } else if (flags & F_MUL_PROD) {
                    load_saved<NT, Y>(w, sv, es);
                    t_mul<NT, X, Y>();
                    eX[0] += es[0];
                } else {
                    load_saved<NT, X>(w, sv, es);
                    eX[0] = es[0];
                }
            }
        } else {  // OP_CONST: childless non-leaf node
            t_const_seq<X>(arg, w.lane >> 4, a.n_states, std::make_integer_sequence<int, 16>{});
            if constexpr (NT == 2) t_const_seq<X + 32>(arg, w.lane >> 4, a.n_states, std::make_integer_sequence<int, 16>{});
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) eX[nt] = 0;
        }
        if (flags & F_RESCALE) t_normalise<NT, X>(eX);
        PROF_T(7)
        if (flags & F_PUSH) push<NT, X>(w, slot, eX);
        PROF_T(8)
        if constexpr (PRE) {
            if (flags & F_PRE_LATE) {
                issue_columns<NT, G>(w, pre_mat);
                refill_codes1<NT>(w, pre_row2);
            }
        }
    }
    return true;
}

// four terms of the root sum: states 4 (4C + i) + g, i = 0..3, for every pattern group
template <int NT, int X, int C>
__device__ __forceinline__ void root_dot4(const double *pi_g, double (&f)[NT])
{
    double m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = pi_g[4 * (4 * C + i)];
    r_fma_into<X + 2 * (4 * C + 0)>(f[0], m[0]);
    r_fma_into<X + 2 * (4 * C + 1)>(f[0], m[1]);
    r_fma_into<X + 2 * (4 * C + 2)>(f[0], m[2]);
    r_fma_into<X + 2 * (4 * C + 3)>(f[0], m[3]);
    if constexpr (NT == 2) {
        r_fma_into<X + 32 + 2 * (4 * C + 0)>(f[1], m[0]);
        r_fma_into<X + 32 + 2 * (4 * C + 1)>(f[1], m[1]);
        r_fma_into<X + 32 + 2 * (4 * C + 2)>(f[1], m[2]);
        r_fma_into<X + 32 + 2 * (4 * C + 3)>(f[1], m[3]);
    }
}
// root: f = sum_i pi_i L_root[i]   (likelihood.rs:236-243)
template <int NT, int X>
__device__ __forceinline__ void root_sum(const PruneArgs &a, const int (&eX)[NT], size_t out_row, unsigned pat, int g)
{
    // pi in four pieces of four: the walk's per-lane state is still live here when the item has another tile, and sixteen
    // doubles at once would push the compiler past its 64 registers (same multiply-adds in the same order)
    double f_[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) f_[nt] = 0.0;
    root_dot4<NT, X, 0>(a.pi_pad + g, f_);
    root_dot4<NT, X, 1>(a.pi_pad + g, f_);
    root_dot4<NT, X, 2>(a.pi_pad + g, f_);
    root_dot4<NT, X, 3>(a.pi_pad + g, f_);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double f = f_[nt];
        f += __shfl_xor(f, 16);
        f += __shfl_xor(f, 32);
        if (g == 0) {
            a.out_f[out_row + pat + nt * 16] = f;
            a.out_e[out_row + pat + nt * 16] = eX[nt];
        }
    }
}

// Merged clade launch: thread 0 of a workgroup waits until the programs listed in `prog`'s dependency record have announced
// all their tiles.  The counters are read with returning atomics (they execute in the XCD's L2, where the announcing
// workgroups -- same XCD, by the host's construction -- added to them after their table stores were acknowledged), then the
// CU's vector L1 is invalidated so that no gather of this item can hit a line from before.  The wait is bounded (a quarter
// of a million polls, far beyond any launch): whoever gives up raises PruneArgs::fault and the evaluation returns NaN
// rather than hanging the device.  The address is in VGPRs: nothing here reads an SGPR written by a VALU instruction.
constexpr int XCC_ID_REG = 63508;   // s_getreg operand: hardware register 20 (XCC_ID), 32 bits from bit 0
// (sync[1]: how many gave up; sync[2]: one of them, block << 8 | XCC.  CRAB_DEBUG_SYNC=1 prints them.)
__device__ __forceinline__ void raise_fault(const PruneArgs &a)
{
    __builtin_nontemporal_store(1, a.sync);
    atomicAdd(a.sync + 1, 1);
    __builtin_nontemporal_store((int)(blockIdx.x << 8 | (__builtin_amdgcn_s_getreg(XCC_ID_REG) & 15u)), a.sync + 2);
}
__device__ __forceinline__ void wait_for_tables(const PruneArgs &a, int prog)
{
    if (threadIdx.x == 0) {
        const int32_t *d = a.deps + (size_t)prog * 16;
        const int n = d[0] < 6 ? d[0] : 6;
        for (int k = 0; k < n; ++k) {
            int32_t *ctr = a.sync + 16 + d[1 + k];
            const int need = d[7 + k];
            int got = 0;
            for (int spin = 0; spin < (1 << 18); ++spin) {
                asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(ctr), "v"(0) : "memory");
                if (got >= need) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (got < need) raise_fault(a);
        }
        asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
}
__device__ __forceinline__ void announce_tiles(const PruneArgs &a, int prog, int tiles)
{
    if (threadIdx.x == 0) {
        int32_t *ctr = a.sync + 16 + prog;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(ctr), "v"(tiles) : "memory");
    }
}

// One workgroup = 4 waves = 64*NT patterns of one site class; each wave walks the whole tree program for its 16*NT
// patterns with two register tiles (see walk()).  The program is per class (matrix indices resolved on the host),
// so the only dependent global reads on the walk are the leaf-edge column gathers.
template <int NT>
__device__ __forceinline__ void prune_body(const PruneArgs &a, long long n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];  // two 32 KiB operand images
    constexpr int X0 = Plan<NT>::T0, X1 = Plan<NT>::T1;
    // the highest fixed register, named once so that the kernel descriptor allocates the whole plan
    asm volatile("" ::: "v255");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const long long n_work = n_tiles * a.K;   // a.K programs are walked by this launch
    constexpr size_t slot_bytes = (size_t)NT * 16 * 64 * 8;

    // Work items are ordered class-major and cut into one contiguous chunk per XCD (workgroups are dealt round-robin
    // to the 8 XCDs), so the workgroups sharing an L2 walk the same class's matrices at about the same time.
    // Placement only matters for speed: any mapping computes every item exactly once.
#ifdef CRAB_PROF
    unsigned prof_acc = 0;
#endif
#ifdef CRAB_CLOCK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (a merged clade launch hands out XCD x's chunk to the workgroups that really run on XCD x: which XCD the device deals
    // workgroup b to depends on what was launched before -- measured: workgroup 0 follows where the previous launch's last one went)
    const int xcd = a.deps ? (int)(__builtin_amdgcn_s_getreg(XCC_ID_REG) & 7u) : (int)(blockIdx.x % N_XCD);
    const int xj = blockIdx.x / N_XCD, per_xcd = gridDim.x / N_XCD;
    const long long chunk_lo = a.cls_list ? a.xcd_bounds[xcd] : n_work * xcd / N_XCD;
    const long long chunk_hi = a.cls_list ? a.xcd_bounds[xcd + 1] : n_work * (xcd + 1) / N_XCD;
    // Work distribution.  With work_ctr: the items of an XCD's chunk are handed out in order, one at a time, to whichever of
    // that XCD's workgroups asks next (the host puts the long items first: the tail of a launch is one short item).  Thread 0
    // takes the next number with a returning atomic at the end of the current item's prologue -- it comes back with the
    // prologue's own loads -- and posts it in an LDS mailbox (two slots used in turn, behind the operand images) that the whole
    // workgroup reads after the barrier that opens the next item.  Which workgroup walks an item matters to no result.  The
    // atomic and the mailbox are inline asm on purpose: with a C++ atomic and a volatile LDS array in the loop the compiler
    // took the whole walk for divergent, ran it under execution masks and spilled forty registers (and that build faulted).
    // Without work_ctr: a chunk's items are dealt round-robin to its workgroups.
    const bool dynamic = a.work_ctr != nullptr;
    const unsigned mbox = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)(2 * P_IMAGE * sizeof(double));
    int *const ctr = uniform_ptr(a.work_ctr + xcd);
    // (with an item list the item's 64-byte record follows its number into the mailbox by LDS-DMA -- four lanes, 16 bytes each --
    // so that the next item opens with an LDS read instead of a scalar load from memory: one round trip less on every item's
    // critical path.  `rec`: 0 / 1, the mailbox half.)
#ifdef CRAB_NO_RECORD_PREFETCH   // (A/B timing)
    const bool records = false;
#else
    const bool records = dynamic && a.items != nullptr;
#endif
    auto post_next_number = [&](int half) {
        if (wave == 0) {
            int number = -1;
            if (lane == 0)
                asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)\n\tds_write_b32 %4, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(number)
                             : "v"(0), "v"(1), "s"(ctr), "v"(mbox + 4u * (unsigned)half)
                             : "memory");
            if (records) {
                const int n0 = __builtin_amdgcn_readfirstlane(number);
                if (n0 >= 0 && n0 < chunk_hi - chunk_lo && lane < 4)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char *)(a.items + chunk_lo + n0) + lane * 16),
                                                     LDS_PTR(smem + 2 * P_IMAGE * sizeof(double) + 64 + 64 * half), 16, 0, 0);
            }
        }
    };
    auto read_number = [&](unsigned slot_addr) {
        int number;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(number) : "v"(slot_addr) : "memory");
        return __builtin_amdgcn_readfirstlane(number);
    };
    // A launch that reads the tables of a merged clade launch checks first that they are all there (every wave for itself: the
    // per-XCD totals and the flag were written back when that kernel ended): an XCD whose chunk found no workgroup, or a
    // workgroup that gave up waiting, must show as NaN and not as a plausible number.
    bool broken = false;
    if (a.expect_tiles > 0) {
        int bad = 0;
        if (lane == 0) {
            int sum = 0;
            for (int x = 0; x < N_XCD; ++x) sum += a.sync[4 + x];
            bad = (a.sync[0] != 0) | (sum != a.expect_tiles);
        }
        broken = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
        if (broken && a.host_flag && threadIdx.x == 0) __builtin_nontemporal_store(1, a.host_flag);
    }
    int announced = 0;   // tiles this workgroup has announced (merged clade launch): added to its XCD's total at the end
    if (dynamic) post_next_number(0);
    if (records && wave == 0) r_wait_vm0();   // (the first record: in the mailbox before the barrier below)
    long long taken = 0;   // (an exit every wave reaches whatever the mailbox holds: never more items than the chunk has)
    long long wi = chunk_lo + xj - per_xcd;
    for (int par = 0;; par ^= 1) {
        __syncthreads();  // the mailbox is written; every wave is done reading LDS for the previous work item
        if (dynamic) {
            wi = chunk_lo + read_number(mbox + 4 * par);
            if (wi < chunk_lo || ++taken > chunk_hi - chunk_lo) break;
        } else {
            wi += per_xcd;
        }
        if (wi >= chunk_hi) break;
        struct {
            int cls;
            long long tile;
            ProgHdr h;
        } item;
        int item_tiles = 1;
        if (a.items) {   // the program's header, the tile and the number of tiles: from the mailbox, or one scalar load
            item.h = records ? read_hdr_lds(mbox + 64u + 64u * (unsigned)par) : load_hdr(a.items + wi);
            item.cls = item.h.prog, item.tile = item.h.tile0;
            item_tiles = __builtin_amdgcn_readfirstlane(item.h.item_tiles);   // (provably wave-uniform: it bounds the tile loop)
        } else {
            const int ci = (int)(wi / n_tiles);
            item.tile = wi - (long long)ci * n_tiles;
            item.cls = a.cls_list ? CONST_PTR(int32_t, a.cls_list)[ci] : ci;
            if (a.hdr) {
                item.h = load_hdr(a.hdr + item.cls);
            } else {
                item.h.n_ops = a.n_ops, item.h.first_gemm_op = a.first_gemm_op, item.h.first_leaf_row = a.first_leaf_row;
                item.h.first_leaf_op = a.first_leaf_op, item.h.second_leaf_row = a.second_leaf_row, item.h.third_leaf_row = a.third_leaf_row;
                item.h.save_cls = item.cls, item.h.no_root = 0, item.h.alt_codes = 0, item.h.first_gemm_mat = -2, item.h.first_leaf_mat = -2;
                item.h.second_leaf_mat = a.second_leaf_op >= 0 ? -2 : -1, item.h.fourth_leaf_row = a.fourth_leaf_row;
            }
        }
        const long long tile = item.tile;
        const int cls = item.cls;
        if (a.deps && (item.h.alt_codes & 2)) {   // (the header is wave-uniform: every wave takes the same side)
            wait_for_tables(a, cls);
            __syncthreads();
        }
        int n_ops, first_gemm_op, first_leaf_row, first_leaf_op, second_leaf_row, third_leaf_row, save_cls, no_root, alt_codes;
        {
            const ProgHdr &h = item.h;
            no_root = h.no_root;
            alt_codes = h.alt_codes;
            n_ops = h.n_ops, first_gemm_op = h.first_gemm_op, first_leaf_row = h.first_leaf_row, first_leaf_op = h.first_leaf_op;
            second_leaf_row = h.second_leaf_row, third_leaf_row = h.third_leaf_row, save_cls = h.save_cls;
        }
        // (-2: no header, the matrices are read from the ops)
        const int first_gemm_mat = item.h.first_gemm_mat, first_leaf_mat = item.h.first_leaf_mat;
        const int second_leaf_mat = item.h.second_leaf_mat, fourth_leaf_row = item.h.fourth_leaf_row;
        const unsigned pat0 = (unsigned)((tile * 4 + wave) * (16 * NT)) + (unsigned)c;  // this lane's pattern in group 0

        Walk<NT> w{a};
        w.ops = a.ops + (size_t)cls * a.n_ops_pad;
        w.ip = 0;
        w.buf = 0;
        w.codes = (alt_codes & 1) ? a.codes_alt + (size_t)(alt_codes >> 2) * (size_t)a.n_code_rows * (size_t)a.n_pat_pad : a.codes;
        w.pat = pat0;
        w.table_off = ((size_t)tile * 4 + wave) * 8192;
        w.gather0 = (const char *)(a.P + P_IMAGE);
        w.stack = (char *)a.stack + ((size_t)blockIdx.x * 4 + wave) * a.max_depth * slot_bytes;
        w.stack_e = a.stack_e + ((size_t)blockIdx.x * 4 + wave) * a.max_depth * (NT * 64);
        w.stack_voff = (unsigned)(lane & 15) * 512u + (unsigned)(lane >> 4) * 16u;
        w.lds_a = (unsigned)(size_t)LDS_PTR(smem) + (g * 4 + (lane & 3)) * 16;
        w.smem = smem;
        w.wave = wave;
        w.lane = lane;
        w.n_ops = n_ops;
        w.save_stride = (size_t)a.save_K * (size_t)n_tiles * 4 * SAVE_TILE_BYTES;
        w.save0 = a.save ? a.save + (((size_t)save_cls * (size_t)n_tiles + (size_t)tile) * 4 + wave) * SAVE_TILE_BYTES : nullptr;
        w.more = item_tiles > 1;

#ifdef CRAB_PROF
        w.prof_acc = prof_acc;
        w.prof_last = __builtin_amdgcn_s_memtime();
#endif
        int e0[NT], e1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            e0[nt] = e1[nt] = 0;
#ifdef CRAB_ABLATE_COALESCE
            w.code_off[nt] = w.code_off2[nt] = (unsigned)lane * 16u;
#else
            w.code_off[nt] = w.code_off2[nt] = (unsigned)a.n_states * 512u + (unsigned)g * 16u;
#endif
        }

        __builtin_amdgcn_s_setprio(CRAB_PRIO_OTHER);
#ifndef CRAB_ABLATE_DMA
        if (first_gemm_op >= 0)
#else
        if (false)
#endif
            dma_matrix(a.P + (size_t)(first_gemm_mat >= 0 ? first_gemm_mat : CONST_PTR(DevOp, w.ops)[first_gemm_op].mat) * P_ELEMS, smem, wave, lane);
        // the program is read 64 ops at a time, one op per lane: no scalar-cache miss sits on the walk's critical path
        w.batch = w.ops[lane];
        if (first_leaf_row >= 0) load_code_offsets<NT>(w, first_leaf_row, w.code_off);
        if constexpr (Plan<NT>::HAS_PRE) {
            if (second_leaf_row >= 0) load_code_offsets<NT>(w, second_leaf_row, w.code_off2);
            if (first_leaf_op >= 0) {
                issue_columns<NT, Plan<NT>::G>(w, first_leaf_mat >= 0 ? first_leaf_mat : CONST_PTR(DevOp, w.ops)[first_leaf_op].mat);
                refill_codes1<NT>(w, third_leaf_row);
                if (second_leaf_mat != -1) {   // op 1 is the same node's second leaf edge: into the landing tile, in the same round trip
                    issue_columns<NT, X1>(w, second_leaf_mat >= 0 ? second_leaf_mat : CONST_PTR(DevOp, w.ops)[1].mat);
                    refill_codes1<NT>(w, fourth_leaf_row);
                }
            }
        }

        if (dynamic) post_next_number(par ^ 1);
        const size_t out_row = (size_t)cls * a.n_pat_pad;
        PROF_T(0)
        // The two tiles swap roles at every product.  A later tile of the item starts in the orientation its predecessor ended
        // in: the last product left the second leaf edge's columns in the tile that is the landing tile from there on.
        // (Two nested loops, one per orientation, and forward jumps out of them: a flag was taken for divergent by the compiler,
        // which then ran both orientations under execution masks; jumps INTO a loop make the control flow irreducible, which
        // the divergence analysis answers by calling everything divergent.)
        int tt = 0;
        // The item's next tile (only programs with F_NEXT_TILE come in items of several tiles: the host's promise).  The last
        // product of the program has requested the columns of this tile's first leaf edges and staged the first product's
        // operand image into the buffer the walk turns to next; what is left of a prologue is the code queue behind those.
        auto next_tile = [&]() -> bool {
            // (readfirstlane: the compiler otherwise takes the tile counter, and through it every request of a cyclic program's
            // last product, for divergent)
            tt = __builtin_amdgcn_readfirstlane(tt + 1);
            if (tt >= item_tiles) return false;
            PROF_T(10)
            w.pat += 64 * NT;
            w.table_off += 4 * 8192;
            if (w.save0) w.save0 += 4 * SAVE_TILE_BYTES;
            w.ip = 0;
            w.more = __builtin_amdgcn_readfirstlane(tt + 1 < item_tiles) != 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) e0[nt] = e1[nt] = 0;
            if (w.n_ops > 64) w.batch = w.ops[lane];
            const ProgHdr h2 = load_hdr(a.items + wi);   // (scalar cache: the item's own record was read into it or lies in L2)
            if (h2.second_leaf_mat != -1)
                refill_codes2<NT>(w, h2.third_leaf_row, h2.fourth_leaf_row);
            else
                refill_codes2<NT>(w, h2.second_leaf_row, h2.third_leaf_row);
            return true;
        };
        for (;;) {   // orientation 0: tile 0 is the current tile
            if (walk<NT, X0, X1>(w, e0, e1)) {
                // (a program that ends in a site likelihood is the last tile of its item whatever the item says: the walk's
                // state is then dead across the root sum, which needs the registers)
                if (!no_root) {
                    root_sum<NT, X0>(a, e0, out_row, w.pat, g);
                    goto item_done;
                }
                if (!next_tile()) goto item_done;
                continue;
            }
            for (;;) {   // orientation 1, until the next product
                if (!walk<NT, X1, X0>(w, e1, e0)) break;
                if (!no_root) {
                    root_sum<NT, X1>(a, e1, out_row, w.pat, g);
                    goto item_done;
                }
                if (!next_tile()) goto item_done;
            }
        }
    item_done:;
        if (records && wave == 0) r_wait_vm0();   // the next item's record has landed in the mailbox
        if (a.deps) {   // a clade program's table tiles: announced once every wave has seen its stores acknowledged
            r_wait_vm0();
            __syncthreads();
            announce_tiles(a, __builtin_amdgcn_readfirstlane(cls), item_tiles);
            announced += item_tiles;
        } else if (broken && !no_root) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a.out_f[out_row + w.pat + nt * 16] = __builtin_nan("");
        }
#ifdef CRAB_PROF
        PROF_T(0)
        prof_acc = w.prof_acc;
#endif
    }
    if (a.deps && announced > 0 && threadIdx.x == 0) {
        int32_t *total = a.sync + 4 + xcd;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(total), "v"(announced) : "memory");
    }
#ifdef CRAB_PROF
    if (a.prof && lane < 2 * PROF_N) ((unsigned *)a.prof)[((size_t)blockIdx.x * 4 + wave) * (2 * PROF_N) + lane] = prof_acc;
#endif
#ifdef CRAB_CLOCK
    if (lane == 0) {
        unsigned long long *cb = g_clock_buf + ((size_t)blockIdx.x * 4 + wave) * 4;
        cb[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        cb[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        cb[2] = clk_r0;
        cb[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);  // XCC_ID, HW_ID
    }
#endif
}

// amdgpu_num_vgpr wants a literal: one entry point per tile width
// amdgpu_num_vgpr alone confines nothing (measured in round 3: the same code comes out with 40, 48, 64 or no attribute at all,
// and under more pressure the allocator simply goes higher -- rounds 1-2 stayed below 64 by the size of the code alone).
// amdgpu_waves_per_eu does: asking for 8 waves per SIMD caps the allocator at 512 / 8 = 64 registers, and it spills to
// scratch beyond that ("failed to meet occupancy target" is the expected remark: the fixed tiles above make it 2).
#ifdef CRAB_PROF
#define CRAB_CC_NT1 96
#define CRAB_WAVES_NT1 5
#else
#define CRAB_CC_NT1 64
#define CRAB_WAVES_NT1 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(CRAB_CC_NT1), amdgpu_waves_per_eu(CRAB_WAVES_NT1, CRAB_WAVES_NT1))) void prune_kernel_nt1(PruneArgs a, long long n_tiles)
{
    static_assert(Plan<1>::CC == CRAB_CC_NT1, "");
    prune_body<1>(a, n_tiles);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(96), amdgpu_waves_per_eu(5, 5))) void prune_kernel_nt2(PruneArgs a, long long n_tiles)
{
    static_assert(Plan<2>::CC == 96, "");
    prune_body<2>(a, n_tiles);
}

// (function-local statics: initialised once, thread-safely, whichever thread asks first -- the shard workers of a
// multi-device handle ask concurrently)
static int chosen_nt()
{
    static const int nt = [] {
        if (const char *s = std::getenv("CRAB_NT")) {
            const int v = std::atoi(s);
            if (v == 1 || v == 2) return v;
        }
        return 1;
    }();
    return nt;
}
static size_t prune_lds_bytes() { return (size_t)2 * P_IMAGE * sizeof(double) + 256; }   // two operand images + the item mailbox (two numbers, two 64-byte records)
// Resident workgroups per CU (every gfx950 device gives the same answer: the kernel's registers and LDS decide).  First asked
// by prune_configure_device, on the thread that creates the engine and with its device current.
static int blocks_per_cu()
{
    static const int bpc = [] {
        int occ = 0;
        const hipError_t err = chosen_nt() == 1
                                   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prune_kernel_nt1, 256, prune_lds_bytes())
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prune_kernel_nt2, 256, prune_lds_bytes());
        const int want = 2;
        int v = (err == hipSuccess && occ >= 1) ? (occ > want ? want : occ) : 1;
        if (const char *s = std::getenv("CRAB_BLOCKS_PER_CU")) {
            const int w = std::atoi(s);
            if (w >= 1 && w <= v) v = w;
        }
        return v;
    }();
    return bpc;
}

int prune_configure_device()
{
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(prune_kernel_nt1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prune_lds_bytes());
    if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(prune_kernel_nt2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prune_lds_bytes());
    if (err == hipSuccess) (void)blocks_per_cu();
    return (int)err;
}
bool prune_device_has_all_xcds(int device)
{
    int n_xcc = 0;
    return hipDeviceGetAttribute(&n_xcc, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && n_xcc == N_XCD;
}
int prune_tile_width() { return 64 * chosen_nt(); }
int prune_reg_stack_slots() { return chosen_nt() == 1 ? Plan<1>::NSTK : Plan<2>::NSTK; }
bool prune_has_prefetch() { return chosen_nt() == 1 ? Plan<1>::HAS_PRE : Plan<2>::HAS_PRE; }
bool prune_supports_paths() { return chosen_nt() == 1 && PATH_OPS; }

int prune_grid_size(int64_t n_work, int device)
{
    // the engine's device, not the thread's current one: one process may hold engines on several GPUs
    int cus = 256, v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) cus = v;
    const int64_t cap = (int64_t)cus * blocks_per_cu();
    const int64_t want = ((n_work + N_XCD - 1) / N_XCD) * N_XCD;  // a multiple of N_XCD: see prune_body
    return (int)(want < cap ? want : cap);
}

size_t prune_stack_doubles(int grid, int max_depth) { return (size_t)grid * 4 * (max_depth > 0 ? max_depth : 1) * chosen_nt() * 16 * 64; }
size_t prune_stack_ints(int grid, int max_depth) { return (size_t)grid * 4 * (max_depth > 0 ? max_depth : 1) * chosen_nt() * 64; }

void launch_prune(const PruneArgs &args, int grid, hipStream_t stream)
{
    PruneArgs a = args;
    if (grid <= 0 || (a.n_ops <= 0 && !a.hdr)) return;
    const int nt = chosen_nt();
    const long long n_tiles = a.n_pat_pad / (64 * nt);
    const size_t lds = prune_lds_bytes();
#ifdef CRAB_PROF
    // one counter buffer per device, handed to the kernel as an argument (a launch with no buffer writes nothing)
    constexpr size_t prof_bytes = sizeof(unsigned long long) * PROF_N * 4 * 4096;
    static unsigned long long *d_prof_dev[16] = {};
    int prof_device = 0;
    unsigned long long *d_prof = nullptr;
    if (hipGetDevice(&prof_device) == hipSuccess && prof_device >= 0 && prof_device < 16 && grid <= 4096) {
        if (!d_prof_dev[prof_device] && hipMalloc((void **)&d_prof_dev[prof_device], prof_bytes) != hipSuccess) d_prof_dev[prof_device] = nullptr;
        d_prof = d_prof_dev[prof_device];
    }
    if (d_prof && hipMemsetAsync(d_prof, 0, prof_bytes, stream) != hipSuccess) d_prof = nullptr;
    if (!d_prof) fprintf(stderr, "[crab prof] no counter buffer on device %d: this launch is not profiled\n", prof_device);
    a.prof = d_prof;
#endif
#ifdef CRAB_CLOCK
    static unsigned long long *d_clock = nullptr;
    if (!d_clock) {
        (void)hipMalloc((void **)&d_clock, sizeof(unsigned long long) * 4 * 4 * 4096);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clock_buf), &d_clock, sizeof(d_clock));
    }
    (void)hipMemsetAsync(d_clock, 0, sizeof(unsigned long long) * 4 * 4 * 4096, stream);
#endif
    if (nt == 1)
        hipLaunchKernelGGL(prune_kernel_nt1, dim3(grid), dim3(256), lds, stream, a, n_tiles);
    else
        hipLaunchKernelGGL(prune_kernel_nt2, dim3(grid), dim3(256), lds, stream, a, n_tiles);
#ifdef CRAB_CLOCK
    {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 4 * 4);
        (void)hipMemcpy(h.data(), d_clock, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        unsigned long long r_first = ~0ull;
        for (size_t wv = 0; wv < (size_t)grid * 4; ++wv)
            if (h[wv * 4 + 1] > 0) {
                ghz.push_back((double)h[wv * 4] / (double)h[wv * 4 + 1] * 0.1);
                cyc.push_back((double)h[wv * 4]);
                if (h[wv * 4 + 2] < r_first) r_first = h[wv * 4 + 2];
            }
        if (std::getenv("CRAB_CLOCK_DUMP"))   // per workgroup: start and end (us after the first wave's start), XCC, SE/CU from HW_ID
            for (int b = 0; b < grid; ++b) {
                const unsigned long long *r = &h[(size_t)b * 16];
                const unsigned hw = (unsigned)r[3], xcc = (unsigned)(r[3] >> 32) & 15;
                fprintf(stderr, "[crab wg] %4d xcc %u se %u cu %2u  start %8.2f end %8.2f us\n", b, xcc, (hw >> 13) & 7, (hw >> 8) & 15,
                        (double)(r[2] - r_first) * 0.01, (double)(r[2] + r[1] - r_first) * 0.01);
            }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        if (!ghz.empty())
            fprintf(stderr, "[crab clock] NT=%d grid=%d: shader clock median %.3f GHz (min %.3f max %.3f); wave life cycles median %.0f max %.0f\n", nt,
                    grid, ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2], cyc.back());
    }
#endif
#ifdef CRAB_PROF
    if (d_prof) {
        (void)hipStreamSynchronize(stream);
        static const char *names[12] = {"prologue+root", "gemm wait vmcnt", "gemm barrier", "gemm dma issue + mfma", "gemm pop-mul",
                                        "leaf gather issue + wait", "leaf multiply", "normalise", "push", "late column issue + loop", "next tile of the item", "op decode"};
        std::vector<unsigned long long> h((size_t)grid * 4 * 12);
        (void)hipMemcpy(h.data(), d_prof, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double tot[12] = {0}, all = 0;
        for (size_t wv = 0; wv < (size_t)grid * 4; ++wv)
            for (int i = 0; i < 12; ++i) tot[i] += (double)h[wv * 12 + i];
        for (int i = 0; i < 12; ++i) all += tot[i];
        fprintf(stderr, "[crab prof] NT=%d grid=%d: mean cycles per wave %.0f\n", nt, grid, all / (grid * 4.0));
        for (int i = 0; i < 12; ++i) fprintf(stderr, "[crab prof]   %-28s %10.0f  %5.1f %%\n", names[i], tot[i] / (grid * 4.0), 100.0 * tot[i] / all);
    }
#endif
}

// =============================================================================================
// kernel 3: mixture / log / sum
// =============================================================================================
constexpr int MIX_THREADS = 256;
constexpr int MIX_CACHED = 16;   // classes whose log terms mix_kernel keeps in registers between its two passes
constexpr double LN2 = 0.693147180559945309417232121458;

// ln(fs * 2^e), with the reference's clamp max(f, 1e-300) applied in log space when asked
__device__ __forceinline__ double ln_site(double fs, int e, bool clamp)
{
    const double lo = -690.7755278982137;  // ln(1e-300)
    if (!(fs > 0.0)) return clamp ? lo : (fs == 0.0 ? -__builtin_inf() : __builtin_nan(""));
    const double l = log(fs) + (double)e * LN2;
    return (clamp && l < lo) ? lo : l;
}

__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < MIX_THREADS / 64; ++i) s += red[i];
    return s;  // valid on thread 0
}

// One pattern's term of the sum: the class mixture, its log (stored per pattern) and the pattern weight.  `a` already
// points at the group's slices; lp = ln p_k in LDS when K <= MIX_CACHED.
__device__ __forceinline__ double mix_pattern(const MixArgs &a, long long p, const double *lp)
{
    if (p >= a.n_pat) return 0.0;
    double lnf;
    if (a.mode == MIX_LINEAR) {
        // ln sum_k freq_k f_k  (branch_site_likelihood.rs:162-168), evaluated around the largest exponent
        int emax = -2147483647 - 1;
        for (int k = 0; k < a.K; ++k) {
            const int ek = a.e[(size_t)k * a.n_pat_pad + p];
            emax = ek > emax ? ek : emax;
        }
        double s = 0.0;
        for (int k = 0; k < a.K; ++k) {
            const size_t o = (size_t)k * a.n_pat_pad + p;
            const int d = a.e[o] - emax;
            s += a.props[k] * (d < -2000 ? 0.0 : ldexp(a.f[o], d));
        }
        lnf = log(s) + (double)emax * LN2;
    } else {
        // x_k = ln max(f_k, 1e-300) + ln p_k ; max + ln sum exp(x - max)  (likelihood.rs:349-370)
        const bool clamp = a.mode == MIX_LOGSUMEXP_CLAMP;
        double mx = -__builtin_inf();
        double s = 0.0;
        if (a.K <= MIX_CACHED) {
            // the usual case (M8: 11 classes): every x_k is computed once and kept in a register for the second pass
            // (the same expression as below, so the same bits: this kernel's transcendental functions were 18 of a step's
            // 1 100 us at configs[1], on the 39 CUs its grid reaches)
            double xs[MIX_CACHED];
#pragma unroll
            for (int k = 0; k < MIX_CACHED; ++k)
                if (k < a.K) {
                    const size_t o = (size_t)k * a.n_pat_pad + p;
                    xs[k] = ln_site(a.f[o], a.e[o], clamp) + lp[k];
                    mx = xs[k] > mx ? xs[k] : mx;
                }
#pragma unroll
            for (int k = 0; k < MIX_CACHED; ++k)
                if (k < a.K) s += exp(xs[k] - mx);
        } else {
            for (int k = 0; k < a.K; ++k) {
                const size_t o = (size_t)k * a.n_pat_pad + p;
                const double x = ln_site(a.f[o], a.e[o], clamp) + log(a.props[k]);
                mx = x > mx ? x : mx;
            }
            for (int k = 0; k < a.K; ++k) {
                const size_t o = (size_t)k * a.n_pat_pad + p;
                const double x = ln_site(a.f[o], a.e[o], clamp) + log(a.props[k]);
                s += exp(x - mx);
            }
        }
        lnf = mx + log(s);
    }
    a.pat_lnl[p] = lnf;
    const double wgt = a.weights[p];
    return wgt != 0.0 ? wgt * lnf : 0.0;
}

// blockIdx.y = group (one parameter vector of a batched evaluation): classes [y*K, (y+1)*K) with their own proportions
__global__ __launch_bounds__(MIX_THREADS) void mix_kernel(MixArgs a)
{
    __shared__ double red[MIX_THREADS / 64];
    const long long p = (long long)blockIdx.x * MIX_THREADS + threadIdx.x;
    const size_t grp = blockIdx.y;
    a.f += grp * a.K * a.n_pat_pad;
    a.e += grp * a.K * a.n_pat_pad;
    a.props += grp * a.K;
    a.pat_lnl += grp * a.n_pat_pad;
    a.partials += grp * gridDim.x;
    __shared__ double lp[MIX_CACHED];   // ln p_k, once per workgroup
    if (a.mode != MIX_LINEAR && a.K <= MIX_CACHED) {
        if ((int)threadIdx.x < a.K) lp[threadIdx.x] = log(a.props[threadIdx.x]);
        __syncthreads();
    }
    const double s = block_sum(mix_pattern(a, p, lp), red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
}

// Few patterns (configs[0]: 7 taxa x 130 codons is one block of 256): mix_kernel and final_sum_kernel as ONE launch.  One
// workgroup per group walks the nb <= MIX_FUSED_BLOCKS blocks of 256 patterns that mix_kernel's grid would have held, takes
// each block's sum with the same reduction, keeps the partial sums in LDS, and adds them as final_sum_kernel does (thread i
// holds partial i, then the same tree): the same bits, one launch less per evaluation.  The limit is ONE block: with two
// (a 30 x 400 gene) the workgroup's second pass over the classes' logs and exponentials costs more than the launch it saves
// (device section 0.147 against 0.140 ms, profiles/r2_fused_mix_timing.txt); the loop stays for the record of that trial.
constexpr int MIX_FUSED_BLOCKS = 1;
__global__ __launch_bounds__(MIX_THREADS) void mix_final_kernel(MixArgs a, int nb)
{
    __shared__ double red[MIX_THREADS / 64];
    __shared__ double part[MIX_FUSED_BLOCKS];
    const size_t grp = blockIdx.x;
    a.f += grp * a.K * a.n_pat_pad;
    a.e += grp * a.K * a.n_pat_pad;
    a.props += grp * a.K;
    a.pat_lnl += grp * a.n_pat_pad;
    __shared__ double lp[MIX_CACHED];
    if (a.mode != MIX_LINEAR && a.K <= MIX_CACHED) {
        if ((int)threadIdx.x < a.K) lp[threadIdx.x] = log(a.props[threadIdx.x]);
        __syncthreads();
    }
    for (int b = 0; b < nb; ++b) {
        const double s = block_sum(mix_pattern(a, (long long)b * MIX_THREADS + threadIdx.x, lp), red);
        if (threadIdx.x == 0) part[b] = s;
        __syncthreads();   // red is reused by the next block's sum; part[b] is read below
    }
    double v = 0.0;
    if ((int)threadIdx.x < nb) v += part[threadIdx.x];
    const double s = block_sum(v, red);
    if (threadIdx.x == 0) a.lnL[grp] = s;
}

// The log-sum-exp mixture with one (pattern, class) per lane: 16 lanes per pattern, 16 patterns per workgroup.  mix_kernel gives
// a lane one pattern and walks its K classes -- 2 K + 1 transcendental functions one after the other on the 9 947 lanes of
// configs[1]; here a lane computes one log and one exponential, the maximum is taken across the 16 lanes (exact in any
// order), and the exponentials are added in class order 0 .. K-1, as mix_kernel adds them: the per-pattern values are the
// same bits.  The workgroup's 16 weighted terms are summed into one partial (16 times as many as mix_kernel's: the final
// sum adds them in a different grouping, so the TOTAL differs from mix_kernel's in its last bits; it is as deterministic).
constexpr int MIX_WIDE_PATTERNS = MIX_THREADS / 16;
__global__ __launch_bounds__(MIX_THREADS) void mix_wide_kernel(MixArgs a)
{
    __shared__ double red[MIX_THREADS / 64];
    __shared__ double lp[MIX_CACHED];
    const size_t grp = blockIdx.y;
    a.f += grp * a.K * a.n_pat_pad;
    a.e += grp * a.K * a.n_pat_pad;
    a.props += grp * a.K;
    a.pat_lnl += grp * a.n_pat_pad;
    a.partials += grp * gridDim.x;
    if ((int)threadIdx.x < a.K) lp[threadIdx.x] = log(a.props[threadIdx.x]);
    __syncthreads();
    const int k = threadIdx.x & 15;
    const long long p = (long long)blockIdx.x * MIX_WIDE_PATTERNS + (threadIdx.x >> 4);
    const bool clamp = a.mode == MIX_LOGSUMEXP_CLAMP;
    const bool live = p < a.n_pat && k < a.K;
    double x = -__builtin_inf();
    if (live) {
        const size_t o = (size_t)k * a.n_pat_pad + p;
        x = ln_site(a.f[o], a.e[o], clamp) + lp[k];
    }
    double mx = x;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off, 16);
        mx = o > mx ? o : mx;
    }
    const double t = live ? exp(x - mx) : 0.0;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < MIX_CACHED; ++j) {
        const double tj = __shfl(t, j, 16);
        if (j < a.K) s += tj;
    }
    double contrib = 0.0;
    if (k == 0 && p < a.n_pat) {
        const double lnf = mx + log(s);
        a.pat_lnl[p] = lnf;
        const double wgt = a.weights[p];
        contrib = wgt != 0.0 ? wgt * lnf : 0.0;
    }
    const double sum = block_sum(contrib, red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = sum;
}

// fixed-order final sum: one workgroup, strided partial sums then a tree
__global__ __launch_bounds__(MIX_THREADS) void final_sum_kernel(const double *partials, int n, double *out)
{
    __shared__ double red[MIX_THREADS / 64];
    partials += (size_t)blockIdx.x * n;   // one workgroup per group
    out += blockIdx.x;
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += MIX_THREADS) v += partials[i];
    const double s = block_sum(v, red);
    if (threadIdx.x == 0) *out = s;
}

int mix_blocks(int64_t n_pat) { return (int)((n_pat + MIX_THREADS - 1) / MIX_THREADS); }
int mix_partials(int64_t n_pat) { return (int)((n_pat + MIX_WIDE_PATTERNS - 1) / MIX_WIDE_PATTERNS); }   // the most any form writes

void launch_mix(const MixArgs &a, hipStream_t stream)
{
    const int nb = mix_blocks(a.n_pat);
    if (nb <= 0) return;
    const int ng = a.n_groups > 0 ? a.n_groups : 1;
    static const bool no_fused = std::getenv("CRAB_NO_FUSED_MIX") != nullptr;   // A/B and the bit-identity test
    if (nb <= MIX_FUSED_BLOCKS && !no_fused) {
        hipLaunchKernelGGL(mix_final_kernel, dim3(ng), dim3(MIX_THREADS), 0, stream, a, nb);
        return;
    }
    // log-sum-exp mixtures of up to 16 classes: one (pattern, class) per lane (6.5 us less per evaluation at configs[1] and at
    // gene size, nothing at 45 000 patterns: profiles/r2_wide_mix_timing.txt); CRAB_NO_WIDE_MIX=1: one pattern per lane
    static const bool wide = std::getenv("CRAB_NO_WIDE_MIX") == nullptr;
    if (wide && a.mode != MIX_LINEAR && a.K <= MIX_CACHED) {
        const int nw = mix_partials(a.n_pat);
        hipLaunchKernelGGL(mix_wide_kernel, dim3(nw, ng), dim3(MIX_THREADS), 0, stream, a);
        hipLaunchKernelGGL(final_sum_kernel, dim3(ng), dim3(MIX_THREADS), 0, stream, a.partials, nw, a.lnL);
        return;
    }
    hipLaunchKernelGGL(mix_kernel, dim3(nb, ng), dim3(MIX_THREADS), 0, stream, a);
    hipLaunchKernelGGL(final_sum_kernel, dim3(ng), dim3(MIX_THREADS), 0, stream, a.partials, nb, a.lnL);
}

__global__ __launch_bounds__(256) void by_class_kernel(const double *__restrict__ f, const int32_t *__restrict__ e, int K,
                                                       long long n_pat, long long n_pat_pad, int clamp, double *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_pat * K) return;
    const long long p = idx / K;
    const int k = (int)(idx - p * K);
    const size_t o = (size_t)k * n_pat_pad + p;
    out[idx] = ln_site(f[o], e[o], clamp != 0);
}

void launch_by_class(const double *f, const int32_t *e, int K, int64_t n_pat, int64_t n_pat_pad, int clamp, double *out,
                     hipStream_t stream)
{
    const long long total = (long long)n_pat * K;
    if (total <= 0) return;
    hipLaunchKernelGGL(by_class_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, f, e, K, (long long)n_pat,
                       (long long)n_pat_pad, clamp, out);
}

// =============================================================================================
// kernel 4: Bayes empirical Bayes grid integration (analysis/beb.py:575-661)
// =============================================================================================
// stage 1: F[k][p] = exp(x_k - max_k x_k), x_k = ln max(f_k, 1e-300) as compute_site_log_likelihoods_by_class
// returns it (likelihood.rs:328-335; beb.py:581-585)
__global__ __launch_bounds__(256) void beb_scale_kernel(BebArgs a)
{
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_pat) return;
    double mx = -__builtin_inf();
    for (int k = 0; k < a.K; ++k) {
        const size_t o = (size_t)k * a.n_pat_pad + p;
        const double x = ln_site(a.f[o], a.e[o], true);
        mx = x > mx ? x : mx;
    }
    for (int k = 0; k < a.K; ++k) {
        const size_t o = (size_t)k * a.n_pat_pad + p;
        a.F[o] = exp(ln_site(a.f[o], a.e[o], true) - mx);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid on lane 0
}

// stage 2: one wavefront per grid point; lanes stride the patterns (coalesced rows of F), fixed-order reduction.
// lnfX_g = sum_p w_p ln f_p(g), a pattern entering only if f_p(g) > 1e-300 (beb.py:604-607)
__global__ __launch_bounds__(256) void beb_grid_lnf_kernel(BebArgs a)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= a.G) return;
    const double *col[BEB_MAX_MIX];
    double pr[BEB_MAX_MIX];
#pragma unroll
    for (int c = 0; c < BEB_MAX_MIX; ++c) {   // fixed trip count: the tables stay in registers
        const bool on = c < a.n_mix;
        col[c] = a.F + (size_t)(on ? a.cols[(size_t)g * a.n_mix + c] : 0) * a.n_pat_pad;
        pr[c] = on ? a.props[(size_t)g * a.n_mix + c] : 0.0;
    }
    double acc = 0.0;
    for (long long p = lane; p < a.n_pat; p += 64) {
        double fh = 0.0;
#pragma unroll
        for (int c = 0; c < BEB_MAX_MIX; ++c)
            if (c < a.n_mix) fh += pr[c] * col[c][p];
        if (fh > 1e-300) acc += a.weights[p] * log(fh);
    }
    acc = wave_sum(acc);
    if (lane == 0) a.lnfx[g] = acc;
}

// stage 3: one wavefront per pattern; lanes stride the grid.  post[p][c] = sum_g (pr_c F_c / f_p(g)) grid_w[g]
// with the reference's guards (f_p(g) < 1e-300 and class terms < 1e-300 are skipped, beb.py:634-647)
__global__ __launch_bounds__(256) void beb_posterior_kernel(BebArgs a)
{
    __shared__ double Fs[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 4 + wave;
    if (p >= a.n_pat) return;
    if (lane < a.K) Fs[wave][lane] = a.F[(size_t)lane * a.n_pat_pad + p];   // K <= 63: one likelihood per lane
    __builtin_amdgcn_wave_barrier();
    double acc[BEB_MAX_MIX];
#pragma unroll
    for (int c = 0; c < BEB_MAX_MIX; ++c) acc[c] = 0.0;
    for (int g = lane; g < a.G; g += 64) {
        double term[BEB_MAX_MIX], fh = 0.0;
#pragma unroll
        for (int c = 0; c < BEB_MAX_MIX; ++c) {
            term[c] = 0.0;
            if (c < a.n_mix) {
                term[c] = a.props[(size_t)g * a.n_mix + c] * Fs[wave][a.cols[(size_t)g * a.n_mix + c]];
                fh += term[c];
            }
        }
        if (fh < 1e-300) continue;
        const double w = a.grid_w[g];
#pragma unroll
        for (int c = 0; c < BEB_MAX_MIX; ++c)
            if (c < a.n_mix && !(term[c] < 1e-300)) acc[c] += term[c] / fh * w;
    }
#pragma unroll
    for (int c = 0; c < BEB_MAX_MIX; ++c)
        if (c < a.n_mix) {
            const double s = wave_sum(acc[c]);
            if (lane == 0) a.post[(size_t)p * a.n_mix + c] = s;
        }
}

// one thread per pattern; the parameter vectors are added in order
__global__ __launch_bounds__(256) void class_posteriors_kernel(const double *__restrict__ f, const int32_t *__restrict__ e, int n_groups,
                                                               int K, long long n_pat, long long n_pat_pad,
                                                               const double *__restrict__ props, const double *__restrict__ weights,
                                                               double *__restrict__ out)
{
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pat) return;
    double acc[POST_MAX_K];
#pragma unroll
    for (int k = 0; k < POST_MAX_K; ++k) acc[k] = 0.0;
    for (int b = 0; b < n_groups; ++b) {
        double lik[POST_MAX_K], row = 0.0;
#pragma unroll
        for (int k = 0; k < POST_MAX_K; ++k) {
            lik[k] = 0.0;
            if (k < K) {
                const size_t o = ((size_t)b * K + k) * n_pat_pad + p;
                const int ex = e[o];
                double v = ex < -1100 ? 0.0 : ldexp(f[o], ex > 1100 ? 1100 : ex);
                v = v > 1e-300 ? v : 1e-300;   // exp(ln max(f, 1e-300)) (likelihood.rs:328-335, beb.py:718)
                lik[k] = v * props[(size_t)b * K + k];
                row += lik[k];
            }
        }
        row = row > 1e-300 ? row : 1e-300;     // beb.py:724-725
        const double w = weights[b] / row;
#pragma unroll
        for (int k = 0; k < POST_MAX_K; ++k)
            if (k < K) acc[k] += w * lik[k];
    }
#pragma unroll
    for (int k = 0; k < POST_MAX_K; ++k)
        if (k < K) out[(size_t)p * K + k] = acc[k];
}

void launch_class_posteriors(const double *f, const int32_t *e, int n_groups, int K, int64_t n_pat, int64_t n_pat_pad,
                             const double *props, const double *weights, double *out, hipStream_t stream)
{
    if (n_pat <= 0) return;
    hipLaunchKernelGGL(class_posteriors_kernel, dim3((unsigned)((n_pat + 255) / 256)), dim3(256), 0, stream, f, e, n_groups, K,
                       (long long)n_pat, (long long)n_pat_pad, props, weights, out);
}

void launch_beb_scale(const BebArgs &a, hipStream_t stream)
{
    if (a.n_pat <= 0) return;
    hipLaunchKernelGGL(beb_scale_kernel, dim3((unsigned)((a.n_pat + 255) / 256)), dim3(256), 0, stream, a);
}
void launch_beb_grid_lnf(const BebArgs &a, hipStream_t stream)
{
    if (a.G <= 0) return;
    hipLaunchKernelGGL(beb_grid_lnf_kernel, dim3((unsigned)((a.G + 3) / 4)), dim3(256), 0, stream, a);
}
void launch_beb_posteriors(const BebArgs &a, hipStream_t stream)
{
    if (a.n_pat <= 0) return;
    hipLaunchKernelGGL(beb_posterior_kernel, dim3((unsigned)((a.n_pat + 3) / 4)), dim3(256), 0, stream, a);
}

}  // namespace crab
