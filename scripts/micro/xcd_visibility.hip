// Is data written by a workgroup on one XCD visible to a workgroup on another XCD INSIDE a launch, and with which cache
// policy bits?  (The pruning launches keep every table and its counter inside one XCD because the L2s are not coherent with
// each other; this measures what it would take to drop that rule.)  Pairs of workgroups (2p, 2p + 1) -- neighbours in the
// dispatch order, so on different XCDs; their XCC ids are printed -- play ping-pong R times: the producer fills 4 KB with the
// round's number and raises a flag, the consumer waits for the flag, reads the 4 KB and counts words that still hold an
// older round (its L2 has them from the round before), then acknowledges.  Flags and acks are device-scope atomics.
//   build: hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/xcd_visibility scripts/micro/xcd_visibility.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int WORDS = 1024, ROUNDS = 200, SPIN = 1 << 20;

template <int ST, int LD>   // ST: 0 plain stores, 1 sc1, 2 sc0 sc1, 3 plain + L2 write-back;  LD: 0 plain loads, 1 sc1, 2 sc0 sc1, 3 plain after L2 invalidate
__global__ void ping_pong(int *data, int *flag, int *ack, int *stale, int *xcc, int *gave_up)
{
    const int pair = blockIdx.x >> 1, role = blockIdx.x & 1, t = threadIdx.x;
    int *d = data + (size_t)pair * WORDS;
    if (t == 0) xcc[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(63508) & 15u);
    int bad = 0;
    __shared__ int dead;
    if (t == 0) dead = 0;
    __syncthreads();
    for (int r = 1; r <= ROUNDS; ++r) {
        if (role == 0) {
            for (int i = t; i < WORDS; i += blockDim.x) {
                int *p = d + i;
                if (ST == 0 || ST == 3) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(r) : "memory");
                if (ST == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
                if (ST == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ST == 3) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __hip_atomic_store(flag + pair, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ack + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < SPIN) __builtin_amdgcn_s_sleep(4);
                if (spins >= SPIN) atomicAdd(gave_up, 1), dead = 1;
            }
            __syncthreads();
            if (dead) break;   // (every wait is bounded and a workgroup that gave up leaves: the launch always drains)
        } else {
            if (t == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < SPIN) __builtin_amdgcn_s_sleep(4);
                if (spins >= SPIN) atomicAdd(gave_up, 1), dead = 1;
                asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");   // this CU's vector L1
                if (LD == 3) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (dead) break;
            for (int i = t; i < WORDS; i += blockDim.x) {
                const int *p = d + i;
                int v;
                if (LD == 0 || LD == 3) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                if (LD == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                if (LD == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                bad += v != r;
            }
            __syncthreads();
            if (t == 0) __hip_atomic_store(ack + pair, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (role == 1 && bad) atomicAdd(stale + pair, bad);
}

template <int ST, int LD>
void run(const char *what, unsigned data_flags = 0)   // data_flags: hipExtMallocWithFlags flags for the data buffer (0: hipMalloc)
{
    const int pairs = 64;
    int *data, *flag, *ack, *stale, *xcc, *gave_up;
    if (data_flags) {
        if (hipExtMallocWithFlags((void **)&data, sizeof(int) * pairs * WORDS, data_flags) != hipSuccess) {
            printf("%-58s: allocation with flags %#x failed\n", what, data_flags);
            return;
        }
    } else
        hipMalloc(&data, sizeof(int) * pairs * WORDS);
    hipMalloc(&flag, sizeof(int) * pairs);
    hipMalloc(&ack, sizeof(int) * pairs);
    hipMalloc(&stale, sizeof(int) * pairs);
    hipMalloc(&xcc, sizeof(int) * 2 * pairs);
    hipMalloc(&gave_up, sizeof(int));
    hipMemset(data, 0, sizeof(int) * pairs * WORDS);
    hipMemset(flag, 0, sizeof(int) * pairs);
    hipMemset(ack, 0, sizeof(int) * pairs);
    hipMemset(stale, 0, sizeof(int) * pairs);
    hipMemset(gave_up, 0, sizeof(int));
    hipLaunchKernelGGL((ping_pong<ST, LD>), dim3(2 * pairs), dim3(256), 0, 0, data, flag, ack, stale, xcc, gave_up);
    hipError_t err = hipDeviceSynchronize();
    std::vector<int> hs(pairs), hx(2 * pairs);
    int hg = 0;
    hipMemcpy(hs.data(), stale, sizeof(int) * pairs, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), xcc, sizeof(int) * 2 * pairs, hipMemcpyDeviceToHost);
    hipMemcpy(&hg, gave_up, sizeof(int), hipMemcpyDeviceToHost);
    long total = 0;
    int cross = 0;
    for (int p = 0; p < pairs; ++p) total += hs[p], cross += hx[2 * p] != hx[2 * p + 1];
    printf("%-58s: %8ld stale words of %d read (%d of %d pairs on two XCDs, %d waits given up, %s)\n", what, total, pairs * WORDS * ROUNDS, cross,
           pairs, hg, hipGetErrorString(err));
    hipFree(data), hipFree(flag), hipFree(ack), hipFree(stale), hipFree(xcc), hipFree(gave_up);
}

int main()
{
    run<0, 0>("plain stores, plain loads");
    run<1, 0>("sc1 stores, plain loads");
    run<2, 0>("sc0 sc1 stores, plain loads");
    run<0, 1>("plain stores, sc1 loads");
    run<1, 1>("sc1 stores, sc1 loads");
    run<2, 2>("sc0 sc1 stores, sc0 sc1 loads");
    run<3, 0>("plain stores + L2 write-back, plain loads");
    run<3, 3>("plain stores + L2 write-back, L2 invalidate + plain loads");
    run<1, 3>("sc1 stores, L2 invalidate + plain loads");
    run<0, 3>("plain stores, L2 invalidate + plain loads");
    run<0, 0>("plain stores, plain loads, data in UNCACHED memory", hipDeviceMallocUncached);
    run<0, 0>("plain stores, plain loads, data in FINE-GRAINED memory", hipDeviceMallocFinegrained);
    return 0;
}
