// Cost of a device-side grid barrier (atomic counter + agent-scope fences) between phases that exchange
// data through global memory, vs one kernel launch per phase: would fusing the colour launches of the
// smallest levels pay?  B workgroups of 64 threads, K phases; each phase reads its neighbours' values of
// the previous phase and writes its own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void phases(double* p, unsigned* counter, int nb, int K) {
    const int i = blockIdx.x * 64 + threadIdx.x, n = nb * 64;
    for (int k = 0; k < K; ++k) {
        const double a = __builtin_nontemporal_load(&p[(i + 64) % n]), b = __builtin_nontemporal_load(&p[(i + n - 64) % n]);
        p[i] = 0.5 * (a + b) + 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(k + 1) * nb;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
}
__global__ void one(double* p, int nb) {
    const int i = blockIdx.x * 64 + threadIdx.x, n = nb * 64;
    const double a = p[(i + 64) % n], b = p[(i + n - 64) % n];
    __syncthreads();
    p[i] = 0.5 * (a + b) + 1.0;
}
int main() {
    double* d; unsigned* c;
    hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22); hipMalloc(&c, 4);
    hipStream_t s; hipStreamCreate(&s);
    const int K = 200;
    for (int nb : {16, 128, 441}) {
        hipMemsetAsync(c, 0, 4, s);
        hipLaunchKernelGGL(phases, dim3(nb), dim3(64), 0, s, d, c, nb, K); hipStreamSynchronize(s);
        hipMemsetAsync(c, 0, 4, s); hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(phases, dim3(nb), dim3(64), 0, s, d, c, nb, K); hipStreamSynchronize(s);
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("%d workgroups: %.2f us per phase with a grid barrier\n", nb, std::chrono::duration<double, std::micro>(t1 - t0).count() / K);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(one, dim3(nb), dim3(64), 0, s, d, nb);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = std::chrono::high_resolution_clock::now();
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t1 = std::chrono::high_resolution_clock::now();
        printf("%d workgroups: %.2f us per phase with one kernel per phase (hipGraph)\n", nb, std::chrono::duration<double, std::micro>(t1 - t0).count() / K);
    }
    return 0;
}
