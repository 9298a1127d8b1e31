// gfx950 (MI355X / CDNA4) kernels of the codon-likelihood engine.
//
//   build_p_kernel   batched P(t) = U diag(exp(lam t)) U^-1 on fp64 MFMA, written in MFMA operand order
//   prune_kernel     Felsenstein pruning: one workgroup walks the whole tree for a tile of site
//                    patterns and one site class; conditional likelihoods live in registers, every
//                    internal edge is a 64x64 fp64 MFMA product whose accumulator is the next
//                    product's B operand; leaf edges are column gathers; child products and
//                    power-of-two rescaling are fused; P matrices stream through LDS by LDS-DMA
//   mix_kernel       per-pattern class mixture, log, pattern-weighted deterministic sum
//
// fp64 MFMA shape: the pruning products use v_mfma_f64_4x4x4_4b_f64.  Measured on MI355X
// (scripts/micro/): it sustains 64-72 TFLOP/s against 35-49 TFLOP/s for v_mfma_f64_16x16x4_f64, and its
// D lane map equals its B lane map, so products chain through registers.
// Written for gfx950 wave64 only.  No other architecture is supported.
#include "crab_kernels.h"

#include <cmath>
#include <cstdlib>

namespace crab {

typedef double v4d __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// =============================================================================================
// kernel 1: batched transition matrices
// =============================================================================================
// One workgroup (4 waves) per matrix; wave w produces rows i in [16w, 16w+16) of P.
// It computes the transposed tiles D^T[j][i] = sum_k Uinv[k][j] * (U[i][k] e^{lam_k t}) with
// A = Uinv^T (M index = j), B = (U diag)^T (N index = i) on v_mfma_f64_16x16x4_f64: with the f64 C/D map
// (col = lane&15, row = (lane>>4) + 4*reg) lane l / register r of tile jt holds
// P[16w + (l&15)][16jt + 4r + (l>>4)], stored into both images of crab_kernels.h.  (This kernel is ~1 % of an
// evaluation; the stores are 8-byte scatters.)
__global__ __launch_bounds__(256) void build_p_kernel(const double *__restrict__ eig, const int32_t *__restrict__ eig_id,
                                                      const double *__restrict__ t_eff, double *__restrict__ P, int n_states)
{
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int mt = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const double *E = eig + (size_t)eig_id[m] * EIG_STRIDE;
    const double *UT = E + EIG_UT, *Uinv = E + EIG_UINV, *lam = E + EIG_LAM;
    const double t = t_eff[m];

    v4d acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = (v4d){0.0, 0.0, 0.0, 0.0};

#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 4 * s + g;
        const double b = UT[k * NS + 16 * mt + c] * exp(lam[k] * t);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const double a = Uinv[k * NS + 16 * jt + c];
            acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[jt], 0, 0, 0);
        }
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
    const int ns_jt = n_states >> 4, ns_r = (n_states >> 2) & 3, ns_g = n_states & 3;
    double *out = P + (size_t)m * P_ELEMS;
    const int i = 16 * mt + c;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = acc[jt][r];
            if (jt == ns_jt && r == ns_r && g == ns_g) v = rs;
            const int j = 16 * jt + 4 * r + g;
            out[p_operand_index(i, j)] = v;
            out[p_gather_index(i, j)] = v;
        }
}

void launch_build_p(const double *eig, const int32_t *eig_id, const double *t_eff, double *P, int M, int n_states,
                    hipStream_t stream)
{
    if (M <= 0) return;
    hipLaunchKernelGGL(build_p_kernel, dim3(M), dim3(256), 0, stream, eig, eig_id, t_eff, P, n_states);
}

// =============================================================================================
// kernel 2: pruning
// =============================================================================================
// Normalise a [16 states-per-lane x 1 pattern] column group so that its maximum over the 64 states
// lies in [1, 2): multiply by an exact power of two and book the exponent.  The 64 states of one
// pattern sit in 16 registers of the 4 lanes {c, c+16, c+32, c+48}.
__device__ __forceinline__ void normalise(double (&v)[16], int &e)
{
    double m = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = v[i] > m ? v[i] : m;
    double o = __shfl_xor(m, 16);
    m = o > m ? o : m;
    o = __shfl_xor(m, 32);
    m = o > m ? o : m;
    if (m > 0.0 && m < __builtin_inf()) {
        int ex = (int)((__double_as_longlong(m) >> 52) & 0x7ff) - 1023;
        ex = ex < -1022 ? -1022 : (ex > 1022 ? 1022 : ex);
        if (ex != 0) {
            const double sc = __longlong_as_double((long long)(1023 - ex) << 52);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= sc;
            e += ex;
        }
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

typedef double v2d __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256, 4) void prune_kernel(PruneArgs a, long long n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];  // one 32 KiB operand image

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const long long n_work = n_tiles * a.K;

    const size_t slot_doubles = (size_t)NT * 16 * 64;
    double *my_stack = a.stack + ((size_t)blockIdx.x * 4 + wave) * a.max_depth * slot_doubles + lane;
    int32_t *my_stack_e = a.stack_e + ((size_t)blockIdx.x * 4 + wave) * a.max_depth * (NT * 64) + lane;
    // this lane's A operand inside a tile pair: element (k = g, i' = lane&3), two consecutive doubles
    const v2d *A2 = (const v2d *)smem + (g * 4 + (lane & 3));

    for (long long w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int cls = (int)(w / n_tiles);
        const long long tile = w - (long long)cls * n_tiles;
        const long long pat0 = (tile * 4 + wave) * (16 * NT) + c;  // this lane's pattern in group 0
        const int32_t *pidx = a.pidx + (size_t)cls * a.n_nodes;

        double cur[NT][16], prod[NT][16];
        int ecur[NT], eprod[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ecur[nt] = 0;
            eprod[nt] = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) cur[nt][i] = prod[nt][i] = 0.0;
        }

        __syncthreads();  // every wave is done reading LDS for the previous work item
        if (a.first_gemm_node >= 0 && !(a.debug & 8)) dma_matrix(a.P + (size_t)pidx[a.first_gemm_node] * P_ELEMS, smem, wave, lane);

        for (int ip = 0; ip < a.n_ops; ++ip) {
            const TreeOp op = a.ops[ip];
            double t[NT][16];
            int et[NT];

            if (op.kind == OP_GEMM) {
                // the operand image of this edge was requested right after the previous product
                if (!(a.debug & 8)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) t[nt][i] = 0.0;
                if (!(a.debug & 4)) {
#pragma unroll
                    for (int sp = 0; sp < 8; ++sp) {
#pragma unroll
                        for (int ic = 0; ic < 4; ++ic) {
                            v2d av[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) av[q] = A2[((ic * 4 + q) * 8 + sp) * 16];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    t[nt][ic * 4 + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[q][0], cur[nt][2 * sp], t[nt][ic * 4 + q], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    t[nt][ic * 4 + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[q][1], cur[nt][2 * sp + 1], t[nt][ic * 4 + q], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) et[nt] = ecur[nt];
                if (!(a.debug & 8)) {
                    __syncthreads();  // all four waves have read the image: the buffer is free for the next edge
                    if (op.arg >= 0) dma_matrix(a.P + (size_t)pidx[op.arg] * P_ELEMS, smem, wave, lane);
                }
            } else if (op.kind == OP_LEAF) {
                // P[i][code] for the 16 states i = 4*idx + g of this lane: 128 contiguous bytes of the gather
                // image in global memory (L2-resident, shared by every workgroup walking this class)
                const v2d *G2 = (const v2d *)(a.P + (size_t)pidx[op.node] * P_ELEMS + P_IMAGE) + g * 8;
                const uint8_t *row = a.codes + (size_t)op.arg * a.n_pat_pad + pat0;
                if (a.debug & 1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        et[nt] = 0;
#pragma unroll
                        for (int idx = 0; idx < 16; ++idx) t[nt][idx] = 0.5;
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int code = row[nt * 16];
                        const v2d *col = G2 + code * 32;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const v2d v = col[q];
                            t[nt][2 * q] = v[0];
                            t[nt][2 * q + 1] = v[1];
                        }
                        et[nt] = 0;
                    }
                }
            } else {  // OP_CONST: childless non-leaf node
                const double v = op.arg ? 1.0 : 0.0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    et[nt] = 0;
#pragma unroll
                    for (int idx = 0; idx < 16; ++idx) t[nt][idx] = (4 * idx + g < a.n_states) ? v : 0.0;
                }
            }

            if (op.flags & F_MUL_PROD) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    et[nt] += eprod[nt];
#pragma unroll
                    for (int idx = 0; idx < 16; ++idx) t[nt][idx] *= prod[nt][idx];
                }
            }
            if ((op.flags & F_POP_MUL) && !(a.debug & 2)) {
                const double *sp = my_stack + (size_t)op.slot * slot_doubles;
                const int32_t *se = my_stack_e + (size_t)op.slot * (NT * 64);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    et[nt] += se[nt * 64];
#pragma unroll
                    for (int idx = 0; idx < 16; ++idx) t[nt][idx] *= sp[(nt * 16 + idx) * 64];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                eprod[nt] = et[nt];
#pragma unroll
                for (int idx = 0; idx < 16; ++idx) prod[nt][idx] = t[nt][idx];
            }
            if (op.flags & (F_FIN | F_RESCALE)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) normalise(prod[nt], eprod[nt]);
            }
            if ((op.flags & F_PUSH) && !(a.debug & 2)) {
                double *sp = my_stack + (size_t)op.slot * slot_doubles;
                int32_t *se = my_stack_e + (size_t)op.slot * (NT * 64);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    se[nt * 64] = eprod[nt];
#pragma unroll
                    for (int idx = 0; idx < 16; ++idx) sp[(nt * 16 + idx) * 64] = prod[nt][idx];
                }
            }
            if (op.flags & F_FIN) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    ecur[nt] = eprod[nt];
#pragma unroll
                    for (int idx = 0; idx < 16; ++idx) cur[nt][idx] = prod[nt][idx];
                }
            }
        }

        // root: f = sum_i pi_i L_root[i]   (likelihood.rs:236-243)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double f = 0.0;
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) f += a.pi_pad[4 * idx + g] * cur[nt][idx];
            f += __shfl_xor(f, 16);
            f += __shfl_xor(f, 32);
            if (g == 0) {
                const size_t o = (size_t)cls * a.n_pat_pad + pat0 + nt * 16;
                a.out_f[o] = f;
                a.out_e[o] = ecur[nt];
            }
        }
    }
}

static int g_nt = 0;
static int chosen_nt()
{
    if (g_nt == 0) {
        g_nt = 1;
        if (const char *s = std::getenv("CRAB_NT")) {
            const int v = std::atoi(s);
            if (v == 1 || v == 2) g_nt = v;
        }
    }
    return g_nt;
}
constexpr size_t PRUNE_LDS_BYTES = P_IMAGE * sizeof(double);
static int g_blocks_per_cu = 0;
static int blocks_per_cu()
{
    if (g_blocks_per_cu == 0) {
        int occ = 0;
        const hipError_t err = chosen_nt() == 1
                                   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prune_kernel<1>, 256, PRUNE_LDS_BYTES)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prune_kernel<2>, 256, PRUNE_LDS_BYTES);
        g_blocks_per_cu = (err == hipSuccess && occ >= 1) ? (occ > 4 ? 4 : occ) : 1;
        if (const char *s = std::getenv("CRAB_BLOCKS_PER_CU")) {
            const int v = std::atoi(s);
            if (v >= 1 && v <= g_blocks_per_cu) g_blocks_per_cu = v;
        }
    }
    return g_blocks_per_cu;
}

int prune_tile_width() { return 64 * chosen_nt(); }

int prune_grid_size(int64_t n_work)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int64_t cap = (int64_t)cus * blocks_per_cu();
    return (int)(n_work < cap ? n_work : cap);
}

size_t prune_stack_doubles(int grid, int max_depth) { return (size_t)grid * 4 * (max_depth > 0 ? max_depth : 1) * chosen_nt() * 16 * 64; }
size_t prune_stack_ints(int grid, int max_depth) { return (size_t)grid * 4 * (max_depth > 0 ? max_depth : 1) * chosen_nt() * 64; }

void launch_prune(const PruneArgs &a_in, int grid, hipStream_t stream)
{
    if (grid <= 0 || a_in.n_ops <= 0) return;
    PruneArgs a = a_in;
    if (const char *s = std::getenv("CRAB_DEBUG_SKIP")) a.debug = std::atoi(s);  // timing ablations only: wrong results
    const int nt = chosen_nt();
    const long long n_tiles = a.n_pat_pad / (64 * nt);
    const size_t lds = PRUNE_LDS_BYTES;
    if (nt == 1)
        hipLaunchKernelGGL(prune_kernel<1>, dim3(grid), dim3(256), lds, stream, a, n_tiles);
    else
        hipLaunchKernelGGL(prune_kernel<2>, dim3(grid), dim3(256), lds, stream, a, n_tiles);
}

// =============================================================================================
// kernel 3: mixture / log / sum
// =============================================================================================
constexpr int MIX_THREADS = 256;
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

__global__ __launch_bounds__(MIX_THREADS) void mix_kernel(MixArgs a)
{
    __shared__ double red[MIX_THREADS / 64];
    const long long p = (long long)blockIdx.x * MIX_THREADS + threadIdx.x;
    double contrib = 0.0;
    if (p < a.n_pat) {
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
            for (int k = 0; k < a.K; ++k) {
                const size_t o = (size_t)k * a.n_pat_pad + p;
                const double x = ln_site(a.f[o], a.e[o], clamp) + log(a.props[k]);
                mx = x > mx ? x : mx;
            }
            double s = 0.0;
            for (int k = 0; k < a.K; ++k) {
                const size_t o = (size_t)k * a.n_pat_pad + p;
                const double x = ln_site(a.f[o], a.e[o], clamp) + log(a.props[k]);
                s += exp(x - mx);
            }
            lnf = mx + log(s);
        }
        a.pat_lnl[p] = lnf;
        const double wgt = a.weights[p];
        contrib = wgt != 0.0 ? wgt * lnf : 0.0;
    }
    const double s = block_sum(contrib, red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
}

// fixed-order final sum: one workgroup, strided partial sums then a tree
__global__ __launch_bounds__(MIX_THREADS) void final_sum_kernel(const double *partials, int n, double *out)
{
    __shared__ double red[MIX_THREADS / 64];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += MIX_THREADS) v += partials[i];
    const double s = block_sum(v, red);
    if (threadIdx.x == 0) *out = s;
}

int mix_blocks(int64_t n_pat) { return (int)((n_pat + MIX_THREADS - 1) / MIX_THREADS); }

void launch_mix(const MixArgs &a, hipStream_t stream)
{
    const int nb = mix_blocks(a.n_pat);
    if (nb <= 0) return;
    hipLaunchKernelGGL(mix_kernel, dim3(nb), dim3(MIX_THREADS), 0, stream, a);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(MIX_THREADS), 0, stream, a.partials, nb, a.lnL);
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

}  // namespace crab
