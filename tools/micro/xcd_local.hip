// Dependent passes among workgroups of ONE XCD, priced against a kernel boundary and against the chip-wide protocol of
// p2p_flag.hip.  The eight L2s are not coherent with each other, which is what makes a chip-wide hand-off cost three
// fabric trips (p2p_flag.hip: 4.7-5.1 us per dependent pass, no better than a kernel boundary).  Inside one XCD the L2
// IS the coherence point: payload stored with plain stores (the per-CU L1 writes through), s_waitcnt vmcnt(0), one L2
// atomic; consumers poll and read with sc0 loads (L1 bypassed, L2 hit).  No fences, no sc1.
//
// Which workgroups share an XCD is not assumed: 8 W workgroups are launched, each reads HW_REG_XCC_ID, those on XCD
// `target` draw a ticket (L2 atomic) and the first W of them do the work; everybody else exits.  All waits time out.
//
//   passes<MODE>: W workgroups x 256 threads, K passes; pass k: wait (MODE 0: barrier over all W workgroups through one
//   counter; MODE 1: the 8 neighbour workgroups' epoch words), read the neighbours' slots of the three previous passes
//   (24 x 16 B per lane; the 4-colour pattern), check every word, write the own slot, publish.
//   hipcc --offload-arch=gfx950 -O2 -o xcd_local tools/micro/xcd_local.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32 xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// aux bit 0 = sc0 (bypass the CU's L1), bit 4 = sc1
__device__ __forceinline__ v4u ld_sc0(__amdgpu_buffer_rsrc_t r, u32 off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 1); }
// polling load: volatile asm (a builtin load may be hoisted out of the spin loop)
// POLL 0: sc0 load; 1: invalidate the CU's L1 (buffer_inv sc0), plain load; 2: sc1 load (agent scope: past the L2)
template <int POLL>
__device__ __forceinline__ u32 poll_sc0(const u32* p) {
    u32 v;
    if (POLL == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POLL == 1) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_plain(__amdgpu_buffer_rsrc_t r, u32 off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0); }
#define L2_ATOMIC __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP      // executes in the L2, no sc1
__device__ __forceinline__ u32 hv(u32 w, u32 k, u32 t) { return (w * 2654435761u) ^ (k * 40503u + 17u) ^ (t << 20); }
__device__ __forceinline__ int nbr(int w, int j, int N, int S) {
    const int d[8] = {-1, 1, -S, S, -S - 1, -S + 1, S - 1, S + 1};
    int x = w + d[j];
    return (x < 0 || x >= N) ? -1 : x;
}
// ctl words: 0 ticket, 1 error/timeout, 2 bad words, 3 barrier counter, 4 workers seen, 16.. xcc histogram; flags[w * 16]
template <int MODE, int POLL>
__global__ __launch_bounds__(256) void passes(v4u* slots, u32* flags, u32* ctl, int W, int S, int K, int target) {
    __shared__ int s_w;
    __shared__ int s_ok;
    const u32 xcc = xcc_id();
    if (threadIdx.x == 0) {
        int w = -1;
        if ((int)xcc == target) { w = (int)__hip_atomic_fetch_add(&ctl[0], 1u, L2_ATOMIC); if (w >= W) w = -1; }
        s_w = w; s_ok = 1;
    }
    __syncthreads();
    const int w = s_w, t = threadIdx.x;
    if (w < 0) return;
    __amdgpu_buffer_rsrc_t r = rsrc_of(slots, (u32)W * 16384u);          // slot: [w][4 colours][256 threads] v4u
    u32 bad = 0;
    for (int k = 0; k < K; ++k) {
        if (k > 0) {
            if (MODE == 0) {
                if (t == 0) {
                    const long long t0 = wall_clock64();
                    while (poll_sc0<POLL>(&ctl[3]) < (u32)(W * k)) {
                        if (wall_clock64() - t0 > 400000) { s_ok = 0; __hip_atomic_store(&ctl[1], 1u, L2_ATOMIC); break; }
                    }
                }
            } else if (t < 8) {
                const int x = nbr(w, t, W, S);
                if (x >= 0) {
                    const long long t0 = wall_clock64();
                    while (poll_sc0<POLL>(&flags[x * 16]) < (u32)k) {
                        if (wall_clock64() - t0 > 400000) { s_ok = 0; __hip_atomic_store(&ctl[1], 1u, L2_ATOMIC); break; }
                    }
                }
            }
            __syncthreads();
            if (!s_ok) break;
            if (POLL == 1) asm volatile("buffer_inv sc0" ::: "memory");      // every wave drops its CU's L1 lines
        }
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = nbr(w, j, W, S);
            if (x < 0) continue;
#pragma unroll
            for (int b = 1; b <= 3; ++b) {
                if (k - b < 0) continue;
                v4u v = POLL == 2 ? __builtin_amdgcn_raw_buffer_load_b128(r, (int)((u32)x * 16384u + (u32)((k - b) & 3) * 4096u + t * 16), 0, 16)
                                  : ld_sc0(r, (u32)x * 16384u + (u32)((k - b) & 3) * 4096u + t * 16);
                bad += (v.x != hv((u32)x, (u32)(k - b), (u32)t));
                acc ^= v.y;
            }
        }
        st_plain(r, (u32)w * 16384u + (u32)(k & 3) * 4096u + t * 16, v4u{hv((u32)w, (u32)k, (u32)t), acc, (u32)k, (u32)w});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            if (MODE == 0) __hip_atomic_fetch_add(&ctl[3], 1u, L2_ATOMIC);
            else __hip_atomic_store(&flags[w * 16], (u32)(k + 1), L2_ATOMIC);
        }
    }
    if (bad) atomicAdd(&ctl[2], bad);
    if (t == 0) { atomicAdd(&ctl[4], 1u); atomicAdd(&ctl[16 + xcc], 1u); }
}

// the same passes as one kernel each (all XCDs, plain loads), for the hipGraph comparison
__global__ __launch_bounds__(256) void one_pass(v4u* slots, u32* ctl, int W, int S, int k) {
    const int w = blockIdx.x, t = threadIdx.x;
    u32 bad = 0, acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int x = nbr(w, j, W, S);
        if (x < 0) continue;
#pragma unroll
        for (int b = 1; b <= 3; ++b) {
            if (k - b < 0) continue;
            v4u v = slots[(size_t)x * 1024 + ((k - b) & 3) * 256 + t];
            bad += (v.x != hv((u32)x, (u32)(k - b), (u32)t));
            acc ^= v.y;
        }
    }
    slots[(size_t)w * 1024 + (k & 3) * 256 + t] = v4u{hv((u32)w, (u32)k, (u32)t), acc, (u32)k, (u32)w};
    if (bad) atomicAdd(&ctl[2], bad);
}

int main() {
    u32 *ctl, *flags; v4u* pay;
    hipMalloc(&ctl, 4096); hipMalloc(&flags, 1024 * 64); hipMalloc(&pay, 1 << 24);
    hipStream_t s; hipStreamCreate(&s);
    const int K = 280;
    for (int W : {4, 16, 32, 64, 128}) {
        const int S = W >= 64 ? 8 : 4;
        for (int poll = 0; poll < 3; ++poll) for (int mode = 0; mode < 2; ++mode) {
            double us = 0; u32 h[32];
            for (int rep = 0; rep < 3; ++rep) {
                hipMemsetAsync(ctl, 0, 4096, s); hipMemsetAsync(flags, 0, 1024 * 64, s); hipMemsetAsync(pay, 0, (size_t)W * 16384, s);
                hipStreamSynchronize(s);
                auto t0 = std::chrono::high_resolution_clock::now();
#define GO(M, P) hipLaunchKernelGGL((passes<M, P>), dim3(8 * W), dim3(256), 0, s, pay, flags, ctl, W, S, K, 0)
                if (mode == 0) { if (poll == 0) GO(0, 0); else if (poll == 1) GO(0, 1); else GO(0, 2); }
                else { if (poll == 0) GO(1, 0); else if (poll == 1) GO(1, 1); else GO(1, 2); }
                hipStreamSynchronize(s);
                us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
            }
            hipMemcpy(h, ctl, 128, hipMemcpyDeviceToHost);
            printf("%3d workgroups on XCD 0 (%u worked; per XCC: %u %u %u %u %u %u %u %u), %s, %s: %.2f us per pass (timeout %u, bad words %u)\n", W, h[4],
                   h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23], mode == 0 ? "one counter barrier" : "neighbour epochs  ",
                   poll == 0 ? "sc0 loads           " : poll == 1 ? "buffer_inv sc0+loads" : "sc1 loads           ", us / K, h[1], h[2]);
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipMemsetAsync(ctl, 0, 4096, s); hipMemsetAsync(pay, 0, (size_t)W * 16384, s);
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(one_pass, dim3(W), dim3(256), 0, s, pay, ctl, W, S, k);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        u32 hb[4]; hipMemcpy(hb, ctl, 16, hipMemcpyDeviceToHost);
        double usg = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemsetAsync(pay, 0, (size_t)W * 16384, s); hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            usg = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        printf("%3d workgroups, kernel per pass in a hipGraph: %.2f us per pass (bad words %u)\n", W, usg / K, hb[2]);
    }
    return 0;
}
