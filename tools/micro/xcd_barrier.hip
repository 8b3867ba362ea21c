// Would a persistent kernel that lives on ONE XCD beat one launch per colour on the smallest levels?
// Launch 8 x W workgroups; those whose wave reports XCC_ID == 0 take a ticket (the first W of them become
// workers, everybody else exits).  The workers share one L2, so a barrier and the data exchange between
// phases can stay inside that L2: workgroup-scope atomics (no sc1: executed in the local L2) and loads
// that bypass the vector L1.  Every wait has a timeout (no hang if the placement assumption fails).
//   hipcc --offload-arch=gfx950 -O2 -o xcd_barrier tools/micro/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID
__device__ __forceinline__ double load_l2(const double* p) {     // bypass the per-CU L1 (sc0), hit the XCD's L2
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// ctl[0] tickets, ctl[1] barrier counter, ctl[2] error flag, ctl[3] number of workgroups seen on XCC 0
__global__ void phases(double* p, unsigned* ctl, int W, int K, int scope_agent) {
    __shared__ int slot;
    if (threadIdx.x == 0) {
        slot = -1;
        if (xcc_id() == 0) {
            atomicAdd(&ctl[3], 1u);
            const unsigned t = __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (t < (unsigned)W) slot = (int)t;
        }
    }
    __syncthreads();
    if (slot < 0) return;
    const int i = slot * 64 + threadIdx.x, n = W * 64;
    for (int k = 0; k < K; ++k) {
        const double a = load_l2(&p[(i + 64) % n]), b = load_l2(&p[(i + n - 64) % n]);
        p[i] = 0.5 * (a + b) + 1.0;
        __builtin_amdgcn_s_waitcnt(0);      // the store has left the CU (L1 is write-through)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(k + 1) * W;
            if (scope_agent) {
                __threadfence();
                atomicAdd(&ctl[1], 1u);
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
                    if (wall_clock64() - t0 > 2000000) { ctl[2] = 1; break; }
            } else {
                __hip_atomic_fetch_add(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const long long t0 = wall_clock64();
                // an atomic read-modify-write is always performed in the L2
                while (__hip_atomic_fetch_add(&ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
                    if (wall_clock64() - t0 > 2000000) { ctl[2] = 1; break; }
            }
        }
        __syncthreads();
        if (ctl[2]) return;
    }
}
int main() {
    double* d; unsigned* c;
    hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22); hipMalloc(&c, 16);
    hipStream_t s; hipStreamCreate(&s);
    const int K = 400;
    for (int agent : {0, 1})
        for (int W : {1, 4, 16, 32, 64}) {
            unsigned h[4];
            double us = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemsetAsync(c, 0, 16, s); hipMemsetAsync(d, 0, 1 << 22, s); hipStreamSynchronize(s);
                auto t0 = std::chrono::high_resolution_clock::now();
                hipLaunchKernelGGL(phases, dim3(8 * W), dim3(64), 0, s, d, c, W, K, agent); hipStreamSynchronize(s);
                us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
            }
            hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            double v; hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost);
            printf("%s-scope barrier, %2d workers (of %u workgroups seen on XCC 0, %u tickets): %.2f us per phase, error %u, p[0] = %.6f\n",
                   agent ? "agent" : "L2-local", W, h[3], h[0], us / K, h[2], v);
        }
    return 0;
}
