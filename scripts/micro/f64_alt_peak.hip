// Microbenchmark: v_mfma_f64_4x4x4_4b_f64 and v_fma_f64 rates on gfx950, for comparison with mfma_f64_peak.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k44(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = a0 + (threadIdx.x % 17) * 0.37, b = b0 - (threadIdx.x % 13) * 0.11;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void kfma(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + (threadIdx.x % 17) * 1e-9, b = b0 - (threadIdx.x % 13) * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char *name, K kern, int nacc, double flop_per_instr_per_wave, int blocks_per_cu)
{
    int grid = 256 * blocks_per_cu;
    double *out; (void)hipMalloc(&out, sizeof(double) * grid * 256);
    const int iters = 200000 / nacc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double waves = grid * 4.0;
    printf("%s NACC=%d waves/SIMD=%d: %.3f ms %.2f TFLOP/s\n", name, nacc, blocks_per_cu, ms, waves * iters * nacc * flop_per_instr_per_wave / ms / 1e9);
    (void)hipFree(out);
}
int main()
{
    run("mfma_f64_4x4x4_4b", k44<8>, 8, 512.0, 1);
    run("mfma_f64_4x4x4_4b", k44<8>, 8, 512.0, 2);
    run("mfma_f64_4x4x4_4b", k44<8>, 8, 512.0, 4);
    run("v_fma_f64", kfma<16>, 16, 128.0, 1);
    run("v_fma_f64", kfma<16>, 16, 128.0, 2);
    run("v_fma_f64", kfma<16>, 16, 128.0, 4);
    return 0;
}
