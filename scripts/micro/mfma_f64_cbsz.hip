// Does v_mfma_f64_4x4x4_4b_f64 on gfx950 honour the A-broadcast controls (cbsz / abid)?  With cbsz = 2 the A values of
// block `abid` (lanes with ((lane >> 2) & 3) == abid) should serve all four blocks.  Checks the result against that model
// and times a stream of such MFMAs against plain ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

template <int CBSZ, int ABID>
__global__ void one(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}

template <int CBSZ>
__global__ void rate(double *out, int iters)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, CBSZ, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, CBSZ, CBSZ ? 1 : 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, CBSZ, CBSZ ? 2 : 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, CBSZ, CBSZ ? 3 : 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, CBSZ, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, CBSZ, CBSZ ? 1 : 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, CBSZ, CBSZ ? 2 : 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, CBSZ, CBSZ ? 3 : 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

// A[blk][i][k] in lane 16k + 4blk + i;  B[blk][k][j] in lane 16k + 4blk + j;  D[blk][i][j] in lane 16i + 4blk + j
static double model(const double *a, const double *b, int lane, int src_of_blk[4])
{
    const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
    double s = 0;
    for (int k = 0; k < 4; ++k) s += a[16 * k + 4 * src_of_blk[blk] + i] * b[16 * k + 4 * blk + j];
    return s;
}

template <int CBSZ, int ABID>
static int check(const double *ha, const double *hb, double *da, double *db, double *dd)
{
    double hd[64];
    hipLaunchKernelGGL((one<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int src[4];
    for (int blk = 0; blk < 4; ++blk) src[blk] = CBSZ == 0 ? blk : CBSZ == 2 ? ABID : ((blk & 2) | (ABID & 1));
    int bad = 0, bad_plain = 0;
    int plain[4] = {0, 1, 2, 3};
    for (int l = 0; l < 64; ++l) {
        bad += std::fabs(hd[l] - model(ha, hb, l, src)) > 1e-9;
        bad_plain += std::fabs(hd[l] - model(ha, hb, l, plain)) > 1e-9;
    }
    printf("cbsz %d abid %d: %2d lanes differ from the broadcast model, %2d from the plain model\n", CBSZ, ABID, bad, bad_plain);
    return bad;
}

int main()
{
    double ha[64], hb[64], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) ha[l] = 1.0 + 0.37 * l + 0.01 * l * l, hb[l] = 2.0 - 0.11 * l + 0.003 * l * l;
    (void)hipMalloc(&da, 512), (void)hipMalloc(&db, 512), (void)hipMalloc(&dd, 512);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice), (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    int bad = check<0, 0>(ha, hb, da, db, dd);
    bad += check<2, 0>(ha, hb, da, db, dd) + check<2, 1>(ha, hb, da, db, dd) + check<2, 2>(ha, hb, da, db, dd) + check<2, 3>(ha, hb, da, db, dd);
    check<1, 0>(ha, hb, da, db, dd), check<1, 1>(ha, hb, da, db, dd);
    printf(bad ? "A-BROADCAST NOT AS MODELLED\n" : "A-broadcast (cbsz 2) works as modelled\n");
    // rate: 2048 workgroups x 256 threads, 8 MFMAs x iters each
    double *out;
    const int blocks = 2048, iters = 20000;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL((rate<0>), dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL((rate<2>), dim3(blocks), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 8 * 512;
            if (rep) printf("%s: %.2f ms, %.1f TFLOP/s\n", v ? "cbsz 2, abid 0..3" : "plain", ms, flop / ms * 1e-9);
        }
    return bad != 0;
}
