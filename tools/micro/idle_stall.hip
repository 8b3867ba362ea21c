// After an idle period, does the GPU stall once shortly after load resumes?  (Seen in every solve() as one
// cycle of ~80 ms instead of 11 ms.)  Batches of ~1 ms of dependent kernels, host time per batch.
//   hipcc --offload-arch=gfx950 -O2 -o idle_stall tools/micro/idle_stall.hip && ./idle_stall [idle_ms] [memory_MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_touch(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0000001 + 1e-9;
}
int main(int argc, char** argv) {
    const int idle_ms = argc > 1 ? atoi(argv[1]) : 50;
    const size_t mb = argc > 2 ? atoll(argv[2]) : 256;
    const size_t n = mb * 1024 * 1024 / 8;
    double* p; hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int rep = 0; rep < 5; ++rep) {
        std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
        std::vector<double> t;
        const auto t0 = now();
        for (int b = 0; b < 250; ++b) {
            const auto a = now();
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, s, p, n);
            hipStreamSynchronize(s);
            t.push_back(std::chrono::duration<double, std::milli>(now() - a).count());
        }
        double tot = std::chrono::duration<double, std::milli>(now() - t0).count(), med = t[200];
        printf("rep %d (idle %d ms before): total %.1f ms, typical batch %.2f ms; slow batches:", rep, idle_ms, tot, med);
        double acc = 0;
        for (size_t i = 0; i < t.size(); ++i) { if (t[i] > 3 * med) printf(" [#%zu at %.0f ms: %.1f ms]", i, acc, t[i]); acc += t[i]; }
        printf("\n");
    }
    return 0;
}
