// Latency of DEPENDENT FP64 vector operations (one wave per SIMD, chains of 64 dependent operations per loop iteration, so
// that loop overhead does not matter), and of CH independent chains interleaved in program order.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ void k(double* out, double a, double b, int iters) {
    double x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < 64; ++d) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
                else if (OP == 1) x[i] = x[i] * a;
                else x[i] = x[i] + b;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH, int OP>
void run(const char* name, int wps) {
    double* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 8);
    const int iters = 400;
    hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    hipLaunchKernelGGL((k<CH, OP>), dim3(256 * wps), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
    (void)hipEventRecord(t0);
    hipLaunchKernelGGL((k<CH, OP>), dim3(256 * wps), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
    (void)hipEventRecord(t1); (void)hipEventSynchronize(t1);
    float ms; (void)hipEventElapsedTime(&ms, t0, t1);
    const double steps = (double)iters * 64;      // dependent steps per chain
    printf("%-4s %d chain(s) per wave, %d wave(s)/SIMD: %.2f ns = %.1f clk@2.4GHz per dependent step; per instruction %.2f ns\n", name, CH, wps,
           ms * 1e6 / steps, ms * 1e-3 * 2.4e9 / steps, ms * 1e6 / steps / CH);
    (void)hipFree(out);
}
int main() {
    run<1, 0>("fma", 1); run<2, 0>("fma", 1); run<4, 0>("fma", 1); run<8, 0>("fma", 1);
    run<1, 1>("mul", 1); run<1, 2>("add", 1);
    run<1, 0>("fma", 2); run<2, 0>("fma", 2); run<1, 0>("fma", 4);
    return 0;
}
