// Microbenchmark: do v_mfma_f64_4x4x4_4b_f64 and v_fma_f64 share one fp64 datapath on gfx950, or do their rates add?
// Each wave runs NM independent MFMA chains and NF independent FMA chains interleaved in one loop.
//   hipcc --offload-arch=gfx950 -O3 f64_mixed_peak.hip -o /tmp/mixed && /tmp/mixed
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NM, int NF>
__global__ __launch_bounds__(256) void kmix(double *out, int iters, double a0, double b0)
{
    double acc[NM > 0 ? NM : 1], f[NF > 0 ? NF : 1];
    for (int i = 0; i < NM; ++i) acc[i] = 0;
    for (int i = 0; i < NF; ++i) f[i] = i;
    double a = a0 + (threadIdx.x % 17) * 1e-9, b = b0 - (threadIdx.x % 13) * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i] = __builtin_fma(f[i], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < NM; ++i) s += acc[i];
    for (int i = 0; i < NF; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NF>
void run(int blocks_per_cu)
{
    int grid = 256 * blocks_per_cu;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * grid * 256);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kmix<NM, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kmix<NM, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double waves = grid * 4.0;
    double mf = waves * iters * 4.0 * NM * 512.0, ff = waves * iters * 4.0 * NF * 128.0;
    printf("NM=%d NF=%d waves/SIMD=%d: %.3f ms  mfma %.2f + fma %.2f = %.2f TFLOP/s\n", NM, NF, blocks_per_cu, ms, mf / ms / 1e9,
           ff / ms / 1e9, (mf + ff) / ms / 1e9);
    (void)hipFree(out);
}

int main()
{
    run<8, 0>(2); run<8, 0>(4);
    run<0, 16>(2); run<0, 16>(4);
    run<8, 8>(2); run<8, 8>(4);
    run<8, 16>(2); run<8, 16>(4);
    run<8, 32>(2); run<8, 32>(4);
    run<4, 32>(4);
    return 0;
}
