// Floor of a dependent kernel chain inside a hipGraph on this GPU: N tiny kernels, each reading and
// writing a few global values (so that the kernel-boundary cache maintenance is exercised).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(double* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
int main() {
    double* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    for (int blocks : {1, 16, 128}) {
        const int N = 2000;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, d, blocks * 64);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("graph: %d blocks x 64 threads: %.2f us per kernel\n", blocks, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        t0 = std::chrono::high_resolution_clock::now();
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, d, blocks * 64);
        hipStreamSynchronize(s);
        t1 = std::chrono::high_resolution_clock::now();
        printf("eager: %d blocks x 64 threads: %.2f us per kernel\n", blocks, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
}
