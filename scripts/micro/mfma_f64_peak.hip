// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the practical peak behind the roofline), with NACC
// independent accumulators per wave and W waves per SIMD; reports the shader clock held during the loop
// (s_memtime cycles / s_memrealtime 100 MHz ticks).  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *stamps, int iters, double a0, double b0)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + (threadIdx.x % 17) * 0.37, b = b0 - (threadIdx.x % 13) * 0.11;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int blocks_per_cu, int threads)
{
    int cus = 256;
    int grid = cus * blocks_per_cu;
    double *out; unsigned long long *st;
    (void)hipMalloc(&out, sizeof(double) * grid * threads);
    (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * grid);
    const int iters = 40000 / NACC;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, st, iters, 1.0, 2.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, st, iters, 1.0, 2.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> mhz(grid), cyc(grid);
    for (int i = 0; i < grid; ++i) { mhz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; cyc[i] = (double)h[2 * i]; }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    double waves = (double)grid * threads / 64, wps = waves / (cus * 4);
    double flops = waves * iters * NACC * 2048.0;
    double mfma_per_simd = (double)iters * NACC * wps;
    printf("NACC=%2d waves/SIMD=%.0f: %.3f ms %6.2f TFLOP/s | clock %.0f MHz | %.1f cycles per MFMA per SIMD\n", NACC, wps, ms,
           flops / ms / 1e9, mhz[grid / 2], cyc[grid / 2] / mfma_per_simd);
    (void)hipFree(out); (void)hipFree(st);
}

int main()
{
    run<4>(1, 256); run<16>(1, 256);
    run<4>(2, 256); run<8>(2, 256);
    run<4>(3, 256); run<4>(4, 256); run<2>(4, 256); run<4>(8, 256); run<1>(8, 256);
    return 0;
}
