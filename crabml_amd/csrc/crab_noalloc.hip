// crab_matrix_exponential_noalloc: crab_matrix_exponential for callers that must not allocate device memory
// (tight serving loops, harnesses that forbid hipMalloc).  Not for use WHILE a stream capture is active: the copies are
// synchronous and the launch goes to the null stream.  The caller's current device is restored on return.
//
// crab_matrix_exponential stages the packed eigensystem, the time, the eigensystem index and the 64 KiB output
// images of build_p_kernel in device buffers that it allocates and frees per call.  Here the same kernel runs on a
// file-scope device array instead: 131 616 bytes per GPU, held from the first call on that GPU until the process ends.
// No hipMalloc / hipFree / hipHostMalloc, no torch: synchronous copies between caller memory and the array, one
// launch on the null stream.
//
// Every call on a GPU shares that GPU's array.  Each call overwrites all of what its kernel reads (the eigensystem is
// zero-padded to 64 states on the host, so a smaller matrix after a larger one sees no stale entries) and reads back
// only what its kernel wrote.  Calls from several host threads on one GPU are serialised by a per-device mutex inside
// the library; calls from several *processes* do not share the array (each process has its own copy).
#include "../../include/crabml_amd.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "crab_eigen.h"
#include "crab_kernels.h"

namespace crab {
void set_global_error(const std::string &msg);   // crab_engine.cpp: what crab_last_error(NULL) returns
}

namespace {

using crab::EIG_STRIDE;
using crab::NS;
using crab::P_ELEMS;

// layout of the scratch array, in doubles: [eigensystem][both images of P][t][eigensystem index (int32) + pad]
constexpr size_t SCR_EIG = 0;
constexpr size_t SCR_P = SCR_EIG + EIG_STRIDE;
constexpr size_t SCR_T = SCR_P + P_ELEMS;
constexpr size_t SCR_ID = SCR_T + 1;
constexpr size_t SCR_DOUBLES = SCR_ID + 1;

static __device__ __align__(256) double scratch[SCR_DOUBLES];

constexpr int MAX_DEVICES = 64;
struct PerDevice {
    std::mutex m;            // serialises the calls of this process on one GPU
    double *ptr = nullptr;   // device address of `scratch` on this GPU (each GPU holds its own copy)
};
PerDevice g_dev[MAX_DEVICES];

// Device address of this GPU's copy of the array; resolved once per device, with that device current.
// (`scratch` itself is a host-side stand-in in host code: only this address may reach a kernel or a copy.)
hipError_t scratch_address(int device, double **out)
{
    PerDevice &pd = g_dev[device];
    if (!pd.ptr) {
        void *p = nullptr;
        const hipError_t err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (err != hipSuccess) return err;
        if (!p) return hipErrorInvalidSymbol;
        pd.ptr = static_cast<double *>(p);
    }
    *out = pd.ptr;
    return hipSuccess;
}

int fail(int code, const std::string &msg)
{
    crab::set_global_error(msg);
    return code;
}

}  // namespace

extern "C" int crab_matrix_exponential_noalloc(int device, int n, const double *Q, const double *pi, double t, double *P_out)
{
    if (!Q || !pi || !P_out) return fail(CRAB_ERR_ARG, "crab_matrix_exponential_noalloc: bad arguments");
    if (n < 1 || n > NS - 1)
        return fail(CRAB_ERR_ARG, "crab_matrix_exponential_noalloc: n must be in [1, 63] (the scratch array holds one 64-state "
                                  "eigensystem and one transition matrix), got " + std::to_string(n));
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) {
        (void)hipGetLastError();
        return fail(CRAB_ERR_DEVICE, "no HIP device available: libcrabml_amd needs an AMD GPU (gfx950); there is no CPU fallback");
    }
    if (device < 0 || device >= nd || device >= MAX_DEVICES)
        return fail(CRAB_ERR_ARG, "device ordinal " + std::to_string(device) + " out of range");

    // host staging lives on the stack / in host vectors: no pinned allocation either
    std::vector<double> eig(EIG_STRIDE), Pf(P_ELEMS);
    if (!crab::spectral_decompose(n, Q, pi, eig.data())) return fail(CRAB_ERR_EIGEN, "Eigendecomposition failed");

    std::lock_guard<std::mutex> lk(g_dev[device].m);
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
    struct RestoreDevice {
        int d;
        ~RestoreDevice() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{caller_device == device ? -1 : caller_device};
    hipError_t err = hipSetDevice(device);
    double *scr = nullptr;
    if (err == hipSuccess) err = scratch_address(device, &scr);
    const int32_t id_word[2] = {0, 0};
    if (err == hipSuccess) err = hipMemcpy(scr + SCR_EIG, eig.data(), EIG_STRIDE * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(scr + SCR_T, &t, sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(scr + SCR_ID, id_word, sizeof(id_word), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        // the kernel writes all 8 192 doubles of both images, so nothing of an earlier call survives in them
        crab::launch_build_p(scr + SCR_EIG, reinterpret_cast<const int32_t *>(scr + SCR_ID), scr + SCR_T, scr + SCR_P, 1, n, false, nullptr);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpy(Pf.data(), scr + SCR_P, P_ELEMS * sizeof(double), hipMemcpyDeviceToHost);
    if (err != hipSuccess) return fail(CRAB_ERR_DEVICE, std::string("crab_matrix_exponential_noalloc: ") + hipGetErrorString(err));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) P_out[(size_t)i * n + j] = Pf[crab::p_gather_index(i, j)];
    return CRAB_OK;
}
