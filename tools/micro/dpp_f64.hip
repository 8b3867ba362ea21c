// Issue rate and dependent latency of v_fmac_f64_dpp row_newbcast (64-bit DPP, gfx90a+) against plain v_fmac_f64 and against
// v_mov_b32_dpp + v_fmac_f64, one wave per SIMD; and a check that the broadcast delivers lane k of each 16-lane row.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, double a, int iters) {
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, u = 1e-9 * threadIdx.x, g = 1.0000001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            if (MODE == 0) {            // two accumulators alternating, DPP broadcast
                asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" : "+v"(x0), "+v"(x1) : "v"(u), "v"(g));
            } else if (MODE == 1) {     // plain
                asm volatile("v_fmac_f64_e32 %0, %2, %3\nv_fmac_f64_e32 %1, %2, %3\n" : "+v"(x0), "+v"(x1) : "v"(u), "v"(g));
            } else if (MODE == 2) {     // one accumulator, DPP: dependent latency
                asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %0, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" : "+v"(x0), "+v"(x1) : "v"(u), "v"(g));
            } else {                    // DPP source produced by the previous instruction (the chain's hand-over): x0 -> dpp -> x1 -> dpp -> x0
                asm volatile("s_nop 1\nv_fmac_f64_dpp %1, %0, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                             "s_nop 1\nv_fmac_f64_dpp %0, %1, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" : "+v"(x0), "+v"(x1) : "v"(u), "v"(g));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1;
}
__global__ void check(double* out) {
    double u = (double)threadIdx.x, acc = 0.0, g = 1.0;
    asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" : "+v"(acc) : "v"(u), "v"(g));
    double acc2 = 1000.0;
    asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, -%2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" : "+v"(acc2) : "v"(u), "v"(g));
    out[threadIdx.x] = acc; out[64 + threadIdx.x] = acc2;
}
template <int MODE>
void run(const char* name) {
    double* out; (void)hipMalloc(&out, 256 * 256 * 8);
    const int iters = 2000;
    hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, 1.0000001, iters);
    (void)hipEventRecord(t0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, 1.0000001, iters);
    (void)hipEventRecord(t1); (void)hipEventSynchronize(t1);
    float ms; (void)hipEventElapsedTime(&ms, t0, t1);
    const double n = (double)iters * 64;
    printf("%-34s %.2f ns per instruction = %.1f clk @2.4 GHz\n", name, ms * 1e6 / n, ms * 1e-3 * 2.4e9 / n);
    (void)hipFree(out);
}
int main() {
    double* out; (void)hipMalloc(&out, 128 * 8);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out);
    double h[128]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { if (h[l] != (double)((l & ~15) + 2)) ++bad; if (h[64 + l] != 1000.0 - (double)((l & ~15) + 3)) ++bad; }
    printf("row_newbcast check: %d wrong lanes (lane 5 got %.0f, want 2; lane 37 got %.0f, want 34)\n", bad, h[5], h[37]);
    run<0>("fmac_f64_dpp, 2 accumulators");
    run<1>("fmac_f64 plain, 2 accumulators");
    run<2>("fmac_f64_dpp, 1 accumulator");
    run<3>("fmac_f64_dpp chained through DPP");
    return 0;
}
