// FP64 vector throughput on this GPU: independent and dependent v_fma_f64 / v_mul_f64 / v_add_f64 streams at 1, 2, 4 waves per
// SIMD (context for the FP64 co-limit of the sweep kernels).  hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int OP>
__global__ void k(double* out, double a, double b, int iters) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
            else if (OP == 1) x[i] = x[i] * a;
            else x[i] = x[i] + b;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP, int OP>
void run(const char* name, int waves_per_simd) {
    double* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 8);
    const int iters = 20000;
    dim3 grid(256 * waves_per_simd), block(256);        // 256 CUs x (4 SIMDs x waves_per_simd) waves
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    hipLaunchKernelGGL((k<ILP, OP>), grid, block, 0, 0, out, 1.0000001, 1e-9, iters);
    hipEventRecord(t0);
    hipLaunchKernelGGL((k<ILP, OP>), grid, block, 0, 0, out, 1.0000001, 1e-9, iters);
    hipEventRecord(t1); hipEventSynchronize(t1);
    float ms; hipEventElapsedTime(&ms, t0, t1);
    const double ninstr = (double)iters * ILP;                    // per wave
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-6s ILP %d, %d wave(s)/SIMD: %.2f clk per wave instruction (per SIMD: %.2f), %.1f TFLOP/s (fma = 2 flop)\n", name, ILP, waves_per_simd,
           clk / ninstr, clk / ninstr / waves_per_simd, (OP == 0 ? 2.0 : 1.0) * ninstr * 64 * 1024 * waves_per_simd / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>("fma", w); run<4, 0>("fma", w); run<8, 0>("fma", w);
        run<8, 1>("mul", w); run<8, 2>("add", w);
    }
    return 0;
}
