// Discover the lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950 by one-hot probing:
// A = e_la, B = e_lb (one lane each set to 1): prints which D lanes become 1.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *mask)
{
    int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (l == 0) mask[la * 64 + lb] = m;
        }
}
int main()
{
    unsigned long long *dm, hm[4096];
    (void)hipMalloc(&dm, sizeof(hm));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dm);
    (void)hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (hm[la * 64 + lb]) {
                printf("  B%d->D{", lb);
                for (int d = 0; d < 64; ++d) if (hm[la * 64 + lb] >> d & 1) printf("%d ", d);
                printf("}");
            }
        printf("\n");
    }
    return 0;
}
