// Microbenchmark: the pruning kernel's product core (gemm_tile) in isolation -- A operand image resident in LDS,
// conditional likelihoods in registers, R chained products per wave, no tree walk.  Gives the ceiling the whole
// kernel can reach with this core at a given occupancy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I crabml_amd/csrc scripts/micro/gemm_tile_rate.hip -o /tmp/gtr && /tmp/gtr
#include "../../crabml_amd/csrc/crab_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crab;

struct NoRide {
    template <int GI>
    __device__ __forceinline__ void at() const {}
    __device__ __forceinline__ void after() const {}
};

#ifdef CRAB_SKIP15_EMU
// What the border of a 60 + 1 split would add per product (timing only): 16 LDS reads of row/column 60 through the
// A ring, 16 + 15 fp64 multiply-adds per pattern group, and the cross-lane sum and broadcast of state 60.
template <int D, int A, int B>
__device__ __forceinline__ void r_fma64()
{
    asm volatile("v_fma_f64 v[%0:%1], v[%2:%3], v[%4:%5], v[%0:%1]" ::"n"(D), "n"(D + 1), "n"(A), "n"(A + 1), "n"(B), "n"(B + 1));
}
template <int X, int Y, int... I>
__device__ __forceinline__ void border_fma(std::integer_sequence<int, I...>)
{
    constexpr int AV = Plan<2>::AV;
    (r_fma64<Y + 30, AV + 2 * (I & 15), X + 2 * I>(), ...);
    (r_fma64<Y + 2 * (I % 15), AV + 2 * (I & 15), X + 30>(), ...);
}
template <int NT, int X, int Y>
__device__ __forceinline__ void border_emu(unsigned lds_a, double seed)
{
    constexpr int AV = Plan<2>::AV;
    for (int q = 0; q < 2; ++q) {
        r_lds_read128<AV + 0, 0>(lds_a); r_lds_read128<AV + 4, 256>(lds_a); r_lds_read128<AV + 8, 512>(lds_a); r_lds_read128<AV + 12, 768>(lds_a);
        r_lds_read128<AV + 16, 1024>(lds_a); r_lds_read128<AV + 20, 1280>(lds_a); r_lds_read128<AV + 24, 1536>(lds_a); r_lds_read128<AV + 28, 1792>(lds_a);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    border_fma<X, Y>(std::make_integer_sequence<int, 16>{});
    if constexpr (NT == 2) border_fma<X + 32, Y + 32>(std::make_integer_sequence<int, 16>{});
    double s = 0.0;
    r_fma_into<Y + 30>(s, seed);
    auto a = __builtin_amdgcn_permlane16_swap(__double2hiint(s), __double2loint(s), false, false);
    auto b2 = __builtin_amdgcn_permlane32_swap((int)a[0], (int)a[1], false, false);
    if (b2[0] == 0x7fffffff) asm volatile("s_nop 0");   // keep the swaps alive
}
#endif

template <int NT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(96))) void k(double *out, const double *img, int reps, unsigned long long *clk)
{
    constexpr int X = Plan<2>::T0, Y = Plan<2>::T1;   // the NT = 2 plan has room for either width
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 4;
    for (int i = threadIdx.x; i < 4096; i += 256) ((double *)smem)[i] = img[i];
    __syncthreads();
    const unsigned lds_a = (unsigned)(size_t)LDS_PTR(smem) + (g * 4 + (lane & 3)) * 16;
    const double seed = 1.0 + 0.001 * threadIdx.x + 1e-7 * (threadIdx.x * 2654435761u % 1000003u);
    t_scale_seq<X>(0.0, std::make_integer_sequence<int, 16 * NT>{});   // whatever was there: finite or not, timing only
    t_const_seq<X>(1, 0, 64, std::make_integer_sequence<int, 16 * NT>{});
    t_scale_seq<X>(seed, std::make_integer_sequence<int, 16 * NT>{});
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        for (int r = 0; r < reps; r += 2) {   // ping-pong, like the walk
            gemm_tile<NT, X, Y>(lds_a, NoRide{});
#ifdef CRAB_SKIP15_EMU
            border_emu<NT, X, Y>(lds_a, seed);
#endif
            gemm_tile<NT, Y, X>(lds_a, NoRide{});
#ifdef CRAB_SKIP15_EMU
            border_emu<NT, Y, X>(lds_a, seed);
#endif
        }
    } else {
        // MODE 1: the kernel's Ride hooks with nothing to do; MODE 2: with the LDS-DMA of an image riding along
        // (into the upper half of a 64 KiB allocation); MODE 3: plus a barrier per product
        Ride<-1, X> ra;
        Ride<-1, Y> rb;
        const int wave = threadIdx.x >> 6;
        ra.dma_src = rb.dma_src = MODE >= 2 ? (const char *)img + wave * 8192 + lane * 16 : nullptr;
        ra.dma_dst = rb.dma_dst = smem + 32768 + wave * 8192;
        ra.img = rb.img = ra.img_b = rb.img_b = nullptr;
        ra.voff = rb.voff = ra.voff_b = rb.voff_b = 0;
        for (int r = 0; r < reps; r += 2) {
            if (MODE >= 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            gemm_tile<NT, X, Y>(lds_a, ra);
            if (MODE >= 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            gemm_tile<NT, Y, X>(lds_a, rb);
        }
    }
    if (lane == 0) {   // shader clock held under this loop: delta s_memtime / delta s_memrealtime x 100 MHz
        clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    double s = 0.0;
    r_fma_into<X>(s, seed);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NT, int MODE>
void run(int blocks_per_cu)
{
    const int grid = 256 * blocks_per_cu, reps = 400;
    double *out, *img;
    unsigned long long *clk;
    (void)hipMalloc(&clk, sizeof(unsigned long long) * grid * 8);
    (void)hipMalloc(&out, sizeof(double) * grid * 256);
    (void)hipMalloc(&img, sizeof(double) * 4096);
    std::vector<double> h(4096, 1.0 / 64);
    if (getenv("RANDOM_DATA")) {   // operand bits matter: the chip holds a lower clock on random mantissas than on constants
        srand(7);
        for (auto &x : h) x = (0.5 + rand() / (double)RAND_MAX) / 64.0;
    }
    (void)hipMemcpy(img, h.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, MODE>), dim3(grid), dim3(256), 65536, 0, out, img, reps, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, MODE>), dim3(grid), dim3(256), 65536, 0, out, img, reps, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * reps * NT * 256 * 512.0;
    std::vector<unsigned long long> hc((size_t)grid * 8);
    (void)hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < grid * 4; ++i) {
        ghz.push_back(hc[2 * i] / (double)hc[2 * i + 1] * 0.1);
        cyc.push_back((double)hc[2 * i]);
    }
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    printf("gemm_tile NT=%d mode=%d waves/SIMD=%d: %.3f ms  %.2f TFLOP/s (64-padded)  shader clock %.3f GHz  %.2f shader cycles/MFMA/SIMD\n", NT, MODE,
           blocks_per_cu, ms, flop / ms / 1e9, ghz[ghz.size() / 2], cyc[cyc.size() / 2] / ((double)reps * NT * 256) / blocks_per_cu * 1.0);
    (void)hipFree(clk);
    (void)hipFree(out);
    (void)hipFree(img);
}

int main()
{
    run<1, 0>(1); run<1, 0>(2); run<1, 1>(2); run<1, 2>(2); run<1, 3>(2);
    run<2, 0>(1); run<2, 0>(2); run<2, 1>(2); run<2, 2>(2); run<2, 3>(2);
    return 0;
}
