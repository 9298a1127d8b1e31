// Microbenchmark: cost of the leaf-edge column gather (8 x global_load_dwordx4 per lane from a 32 KiB image chosen per
// op, column chosen per pattern) under different image layouts, 8 waves per CU, images L2-resident.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/gather_rate.hip -o /tmp/gr && /tmp/gr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void k(const char *imgs, const unsigned char *codes, double *out, int n_img, int reps)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        const char *img = imgs + (size_t)((r * 7 + blockIdx.x % 3) % n_img) * 32768;
        const unsigned code = codes[((size_t)(wave_global % 64) * 997 + r) * 16 % 65536 + c];
        unsigned off;
        if (V == 0) off = code * 512 + g * 16;          // piece q at +64q  (current layout)
        if (V == 1) off = code * 512 + g * 128;         // piece q at +16q  (lane-contiguous 128 B)
        if (V == 2) off = lane * 16;                    // piece q at +1024q (fully coalesced, no gather)
        if (V == 3) off = (code & 0) * 512 + g * 16;    // every pattern the same column
        if (V == 4) off = codes[((size_t)(wave_global % 64) * 997 + r) * 16 % 65536 + (lane >> 2)] * 512 + (lane & 3) * 16;  // quad = one pattern
        v2d v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *(const v2d *)(img + off + q * (V == 0 || V == 3 || V == 4 ? 64 : V == 1 ? 16 : 1024));
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q][0] + v[q][1];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
void run(const char *name, const char *imgs, const unsigned char *codes, double *out, int n_img)
{
    const int grid = 512, reps = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, imgs, codes, out, n_img, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, imgs, codes, out, n_img, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms: %.0f cycles per leaf op per wave (8 waves/CU), %.2f TB/s\n", name, ms, ms * 1e-3 * 2.4e9 / reps,
           (double)grid * 4 * reps * 8192 / ms / 1e9);
}

int main()
{
    const int n_img = 100;
    char *imgs; unsigned char *codes; double *out;
    (void)hipMalloc(&imgs, (size_t)n_img * 32768);
    (void)hipMemset(imgs, 0, (size_t)n_img * 32768);
    std::vector<unsigned char> h(65536 + 64);
    srand(1);
    for (auto &x : h) x = rand() % 62;
    (void)hipMalloc(&codes, h.size());
    (void)hipMemcpy(codes, h.data(), h.size(), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, sizeof(double) * 512 * 256);
    run<0>("piece-interleaved columns (current)", imgs, codes, out, n_img);
    run<1>("lane-contiguous 128 B", imgs, codes, out, n_img);
    run<2>("fully coalesced (reference)", imgs, codes, out, n_img);
    run<3>("one column for all patterns", imgs, codes, out, n_img);
    run<4>("quad of lanes = one 64 B segment", imgs, codes, out, n_img);
    return 0;
}
