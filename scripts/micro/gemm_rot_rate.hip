// Microbenchmark + correctness check of a product core that reads every A block once: "rotated tile layout".
// Register (c, r) of a tile holds, in pattern quad b, state group 4c + ((b - SGN r) & 3).  One 16-byte LDS read then
// carries four different 4x4 blocks of P (one per quad) for two k-groups, and row_ror:4 DPP moves (in place, in the
// shadow of the MFMAs) turn it into the A operands of 8 MFMAs.  32 reads + 384 DPP moves per 256 MFMAs, against the
// kernel's 128 reads (scripts/micro/gemm_tile_rate.hip measures that one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/gemm_rot_rate.hip -o gemm_rot_rate
#include "../../crabml_amd/csrc/crab_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crab;

#ifndef SGN
#define SGN 1
#endif

template <int R>
__device__ __forceinline__ void r_ror4()   // v[R] <- v[R] rotated by one quad within each row of 16 lanes
{
    asm volatile("v_mov_b32_dpp v[%0], v[%0] row_ror:4 row_mask:0xf bank_mask:0xf" ::"n"(R));
}
template <int B>
__device__ __forceinline__ void rot_base()
{
    r_ror4<B>(), r_ror4<B + 1>(), r_ror4<B + 2>(), r_ror4<B + 3>();
}

__host__ __device__ inline int p_rot_index(int row, int col)
{
    const int I = row >> 2, i = row & 3, ic = I >> 2, u = I & 3;
    const int kg = col >> 2, k = col & 3, kc = kg >> 2, v = kg & 3, delta = (v - u) & 3;
    return ((((ic * 2 + (kc >> 1)) * 4 + delta) * 64) + 16 * k + 4 * u + i) * 2 + (kc & 1);
}

// quad QD = (kcp, delta): the four reads (ic = 0..3) sit in base set QD & 1; R = rotation step 0..3
template <int X, int Y, int QD, int R>
__device__ __forceinline__ void rot_step()
{
    constexpr int AV = Plan<2>::AV, kcp = QD >> 2, delta = QD & 3, set = (QD & 1) * 16;
    constexpr int rp = (R - SGN * delta) & 3;
    constexpr bool first = QD == 0;
    // lo halves (k-group 2kcp) of the four reads, then hi halves (2kcp + 1); each base is rotated right after its hi MFMA
    r_mfma<Y + 2 * (0 + R), AV + set + 0, X + 2 * (4 * (2 * kcp) + rp), first>();
    r_mfma<Y + 2 * (4 + R), AV + set + 4, X + 2 * (4 * (2 * kcp) + rp), first>();
    r_mfma<Y + 2 * (8 + R), AV + set + 8, X + 2 * (4 * (2 * kcp) + rp), first>();
    r_mfma<Y + 2 * (12 + R), AV + set + 12, X + 2 * (4 * (2 * kcp) + rp), first>();
    r_mfma<Y + 2 * (0 + R), AV + set + 2, X + 2 * (4 * (2 * kcp + 1) + rp), false>();
    if constexpr (R < 3) rot_base<AV + set + 0>();
    r_mfma<Y + 2 * (4 + R), AV + set + 6, X + 2 * (4 * (2 * kcp + 1) + rp), false>();
    if constexpr (R < 3) rot_base<AV + set + 4>();
    r_mfma<Y + 2 * (8 + R), AV + set + 10, X + 2 * (4 * (2 * kcp + 1) + rp), false>();
    if constexpr (R < 3) rot_base<AV + set + 8>();
    r_mfma<Y + 2 * (12 + R), AV + set + 14, X + 2 * (4 * (2 * kcp + 1) + rp), false>();
    if constexpr (R < 3) rot_base<AV + set + 12>();
    if constexpr (R < 3) rot_step<X, Y, QD, R + 1>();
}
template <int QD>
__device__ __forceinline__ void rot_reads(unsigned lds_addr)
{
    constexpr int AV = Plan<2>::AV, kcp = QD >> 2, delta = QD & 3, set = (QD & 1) * 16;
    r_lds_read128<AV + set + 0, ((0 * 2 + kcp) * 4 + delta) * 1024>(lds_addr);
    r_lds_read128<AV + set + 4, ((1 * 2 + kcp) * 4 + delta) * 1024>(lds_addr);
    r_lds_read128<AV + set + 8, ((2 * 2 + kcp) * 4 + delta) * 1024>(lds_addr);
    r_lds_read128<AV + set + 12, ((3 * 2 + kcp) * 4 + delta) * 1024>(lds_addr);
}
template <int X, int Y, int QD>
__device__ __forceinline__ void rot_quad(unsigned lds_addr)
{
    if constexpr (QD + 1 < 8) {
        rot_reads<QD + 1>(lds_addr);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    rot_step<X, Y, QD, 0>();
    if constexpr (QD + 1 < 8) rot_quad<X, Y, QD + 1>(lds_addr);
}
template <int X, int Y>
__device__ __forceinline__ void gemm_rot(unsigned lds_addr)   // lds_addr: image base + lane * 16
{
    rot_reads<0>(lds_addr);
    rot_quad<X, Y, 0>(lds_addr);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

template <int... I>
__device__ __forceinline__ void tile_in(const double *src, int lane, std::integer_sequence<int, I...>)
{
    (r_set<Plan<2>::T0 + 2 * I>(src[I * 64 + lane]), ...);
}
template <int... I>
__device__ __forceinline__ void tile_out(double *dst, int lane, std::integer_sequence<int, I...>)
{
    double f[16];
    ((f[I] = 0.0, r_fma_into<Plan<2>::T1 + 2 * I>(f[I], 1.0), dst[I * 64 + lane] = f[I]), ...);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(96))) void k(double *out, const double *img, const double *xin, int reps, unsigned long long *clk)
{
    constexpr int X = Plan<2>::T0, Y = Plan<2>::T1;
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) ((double *)smem)[i] = img[i];
    __syncthreads();
    const unsigned lds_a = (unsigned)(size_t)LDS_PTR(smem) + lane * 16;
    tile_in(xin + wave * 1024, lane, std::make_integer_sequence<int, 16>{});
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; r += 2) {
        gemm_rot<X, Y>(lds_a);
        if (reps > 1) gemm_rot<Y, X>(lds_a);
    }
    if (lane == 0) {
        clk[(blockIdx.x * 4 + wave) * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[(blockIdx.x * 4 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (reps == 1) {
        if (blockIdx.x == 0) tile_out(out + wave * 1024, lane, std::make_integer_sequence<int, 16>{});
    } else {
        double s = 0.0;
        r_fma_into<X>(s, 1.0);
        out[4096 + blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main()
{
    std::vector<double> P(4096), img(4096), Xs(64 * 64), xin(4096), yout(4096);
    srand(7);
    for (auto &x : P) x = (0.5 + rand() / (double)RAND_MAX) / 64.0;
    for (auto &x : Xs) x = 0.5 + rand() / (double)RAND_MAX;       // X[state][pattern], 64 patterns = 4 waves x 16
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 64; ++c) img[p_rot_index(r, c)] = P[r * 64 + c];
    auto state_of = [](int reg, int lane) { const int c = reg >> 2, r = reg & 3, blk = (lane >> 2) & 3; return 4 * (4 * c + ((blk - SGN * r) & 3)) + (lane >> 4); };
    for (int w = 0; w < 4; ++w)
        for (int reg = 0; reg < 16; ++reg)
            for (int lane = 0; lane < 64; ++lane) xin[w * 1024 + reg * 64 + lane] = Xs[state_of(reg, lane) * 64 + 16 * w + (lane & 15)];
    double *dout, *dimg, *dx;
    unsigned long long *clk;
    const int grid = 512, reps = 400;
    (void)hipMalloc(&dout, sizeof(double) * (4096 + grid * 256)), (void)hipMalloc(&dimg, 32768), (void)hipMalloc(&dx, 32768);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * grid * 8);
    (void)hipMemcpy(dimg, img.data(), 32768, hipMemcpyHostToDevice), (void)hipMemcpy(dx, xin.data(), 32768, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, dout, dimg, dx, 1, clk);
    (void)hipMemcpy(yout.data(), dout, 32768, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int w = 0; w < 4; ++w)
        for (int reg = 0; reg < 16; ++reg)
            for (int lane = 0; lane < 64; ++lane) {
                const int st = state_of(reg, lane), pat = 16 * w + (lane & 15);
                double ref = 0.0;
                for (int kk = 0; kk < 64; ++kk) ref += P[st * 64 + kk] * Xs[kk * 64 + pat];
                worst = std::max(worst, std::abs(yout[w * 1024 + reg * 64 + lane] - ref) / ref);
            }
    printf("SGN %d: largest relative difference of Y = P X from the host product: %.3e  (%s)\n", SGN, worst, worst < 1e-13 ? "layout and rotations as modelled" : "WRONG");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, dout, dimg, dx, reps, clk);
        (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc((size_t)grid * 8);
    (void)hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < grid * 4; ++i) ghz.push_back(hc[2 * i] / (double)hc[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("gemm_rot, 2 waves/SIMD, random operands: %.3f ms  %.2f TFLOP/s (64-padded)  shader clock %.3f GHz\n", ms,
           (double)grid * 4 * reps * 256 * 512.0 / ms / 1e9, ghz[ghz.size() / 2]);
    return worst < 1e-13 ? 0 : 1;
}
