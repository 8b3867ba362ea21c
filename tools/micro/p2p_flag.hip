// Point-to-point dependency flags between workgroups of ONE launch, priced against a kernel boundary.
// Question: can the colour launches of a coarse-level smoothing call (7 launches of ~7 us, each
// followed by a cold first load) become ONE persistent launch in which a workgroup waits only for the <= 8 workgroups
// that own its neighbouring lines?
//
// Protocol under test (MI355X_MICROARCH.md, inter-workgroup visibility, "sc1" row): payload stored write-through
// (buffer_store_dwordx4 sc1), s_waitcnt vmcnt(0), ONE lane stores the workgroup's epoch word (sc1); a consumer polls
// the epoch words of its neighbours relaxed (sc1 loads) and then reads their payload with sc1 loads only.  No fences.
//
//   A  ping-pong of two workgroups (same XCD / different XCD): one-way latency of a flag, with and without 1 KB payload
//   B  N workgroups, K passes; pass k: wait for the 8 neighbours (w +- 1, w +- S, w +- S +- 1) to have finished pass
//      k-1, read their slots of the three previous passes (the 4-colour pattern: a pass writes slot k % 4 and reads
//      the other three), write the own slot, publish.  Every word is checked.  Compared with the same work as one
//      kernel per pass inside a hipGraph.
//   hipcc --offload-arch=gfx950 -O2 -o p2p_flag tools/micro/p2p_flag.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ u32 xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ v4u ld_sc1(__amdgpu_buffer_rsrc_t r, u32 off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16); }
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, u32 off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 16); }
__device__ __forceinline__ bool wait_ge(u32* flag, u32 want, u32* err) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, RLX_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 400000) { __hip_atomic_store(err, 1u, RLX_AGENT); return false; }   // 4 ms
    }
    return true;
}

// ---- A: ping-pong ----------------------------------------------------------------------------------------------
// ctl[0] flag A->B, ctl[16] flag B->A, ctl[32] error, ctl[33..34] xcc ids, ctl[40..41] ticks (wall_clock64 = 100 MHz)
__global__ void pingpong(u32* ctl, v4u* payload, int partner, int K, int with_payload) {
    const int me = (blockIdx.x == 0) ? 0 : (blockIdx.x == (unsigned)partner ? 1 : -1);
    if (me < 0) return;
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = rsrc_of(payload, 2 * 1024);
    if (lane == 0) ctl[33 + me] = xcc_id();
    long long t0 = 0;
    u32 bad = 0;
    for (int k = 1; k <= K; ++k) {
        if (k == 2 && me == 0) t0 = wall_clock64();
        if (me == 0) {
            if (with_payload) { st_sc1(r, lane * 16, v4u{(u32)k, (u32)lane, 7u, 9u}); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&ctl[0], (u32)k, RLX_AGENT);
            if (lane == 0) wait_ge(&ctl[16], (u32)k, &ctl[32]);
            __builtin_amdgcn_wave_barrier();
            if (with_payload) { v4u v = ld_sc1(r, 1024 + lane * 16); bad += (v.x != (u32)k) | (v.y != (u32)lane); }
        } else {
            if (lane == 0) wait_ge(&ctl[0], (u32)k, &ctl[32]);
            __builtin_amdgcn_wave_barrier();
            if (with_payload) {
                v4u v = ld_sc1(r, lane * 16); bad += (v.x != (u32)k) | (v.y != (u32)lane);
                st_sc1(r, 1024 + lane * 16, v4u{(u32)k, (u32)lane, 1u, 2u});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&ctl[16], (u32)k, RLX_AGENT);
        }
        if (__hip_atomic_load(&ctl[32], RLX_AGENT)) break;
    }
    if (me == 0 && lane == 0) { const long long t1 = wall_clock64(); ctl[40] = (u32)(t1 - t0); }
    if (bad) atomicAdd(&ctl[35], bad);
}

// ---- B: neighbour-only dependencies, the 4-colour access pattern ------------------------------------------------
// slots: [w][c = pass % 4][64 lanes] v4u; flags[w * 16] (one 64-byte line each)
__device__ __forceinline__ u32 hv(u32 w, u32 k, u32 lane) { return (w * 2654435761u) ^ (k * 40503u + 17u) ^ (lane << 24); }
__device__ __forceinline__ int nbr(int w, int j, int N, int S) {
    const int d[8] = {-1, 1, -S, S, -S - 1, -S + 1, S - 1, S + 1};
    int x = w + d[j];
    return (x < 0 || x >= N) ? -1 : x;
}
template <bool PERSIST, bool SPEC = false>
__global__ void passes(v4u* slots, u32* flags, u32* ctl, int N, int S, int K, int k_only) {
    const int w = blockIdx.x, lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = rsrc_of(slots, (u32)N * 4096u);
    u32 bad = 0;
    for (int k = PERSIST ? 0 : k_only; k < (PERSIST ? K : k_only + 1); ++k) {
        if (PERSIST && SPEC) {
            // speculative: the neighbours' epoch words and their payload are requested TOGETHER; when the epochs are
            // already there (the common case once the pipeline runs) the pass costs one round trip, otherwise retry
            const long long t0 = wall_clock64();
            u32 acc = 0, badk = 0;
            for (;;) {
                u32 f = 0xffffffffu;
                if (k > 0 && lane < 8) { const int x = nbr(w, lane, N, S); if (x >= 0) f = __hip_atomic_load(&flags[x * 16], RLX_AGENT); }
                acc = 0; badk = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int x = nbr(w, j, N, S);
                    if (x < 0) continue;
#pragma unroll
                    for (int b = 1; b <= 3; ++b) {
                        if (k - b < 0) continue;
                        v4u v = ld_sc1(r, (u32)x * 4096u + (u32)((k - b) & 3) * 1024u + lane * 16);
                        badk += (v.x != hv((u32)x, (u32)(k - b), (u32)lane));
                        acc ^= v.y;
                    }
                }
                if (__all((int)(f >= (u32)k))) break;
                if (wall_clock64() - t0 > 400000) { __hip_atomic_store(&ctl[0], 1u, RLX_AGENT); break; }
            }
            bad += badk;
            v4u o{hv((u32)w, (u32)k, (u32)lane), acc, (u32)k, (u32)w};
            st_sc1(r, (u32)w * 4096u + (u32)(k & 3) * 1024u + lane * 16, o);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&flags[w * 16], (u32)(k + 1), RLX_AGENT);
            continue;
        }
        if (PERSIST && k > 0) {
            // lanes 0..7 poll one neighbour each
            bool ok = true;
            if (lane < 8) { const int x = nbr(w, lane, N, S); if (x >= 0) ok = wait_ge(&flags[x * 16], (u32)k, &ctl[0]); }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (!ok) break;
        }
        // read the three previous passes' slots of the 8 neighbours (24 x 16 B per lane), check, fold
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = nbr(w, j, N, S);
            if (x < 0) continue;
#pragma unroll
            for (int b = 1; b <= 3; ++b) {
                if (k - b < 0) continue;
                v4u v;
                if (PERSIST) v = ld_sc1(r, (u32)x * 4096u + (u32)((k - b) & 3) * 1024u + lane * 16);
                else v = slots[(size_t)x * 256 + ((k - b) & 3) * 64 + lane];
                bad += (v.x != hv((u32)x, (u32)(k - b), (u32)lane));
                acc ^= v.y;
            }
        }
        v4u o{hv((u32)w, (u32)k, (u32)lane), acc, (u32)k, (u32)w};
        if (PERSIST) {
            st_sc1(r, (u32)w * 4096u + (u32)(k & 3) * 1024u + lane * 16, o);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&flags[w * 16], (u32)(k + 1), RLX_AGENT);
        } else {
            slots[(size_t)w * 256 + (k & 3) * 64 + lane] = o;
        }
    }
    if (bad) atomicAdd(&ctl[1], bad);
}

int main() {
    u32* ctl; v4u* pay;
    hipMalloc(&ctl, 4096); hipMalloc(&pay, 1 << 24);
    hipStream_t s; hipStreamCreate(&s);
    // A
    for (int partner : {8, 1, 3}) for (int wp : {0, 1}) {
        u32 h[64];
        const int K = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemsetAsync(ctl, 0, 4096, s);
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, s, ctl, pay, partner, K, wp);
            hipStreamSynchronize(s);
        }
        hipMemcpy(h, ctl, 256, hipMemcpyDeviceToHost);
        printf("A ping-pong block 0 (xcc %u) <-> block %d (xcc %u), %s: one way %.3f us, error %u, bad words %u\n", h[33], partner, h[34],
               wp ? "1 KB payload each way" : "flag only", h[40] * 0.01 / (K - 1) / 2, h[32], h[35]);
    }
    // B
    u32* flags; hipMalloc(&flags, 1024 * 64);
    const int K = 280;
    for (int N : {16, 64, 128, 256, 512}) {
        const int S = N >= 64 ? 8 : 4;
        u32 h[4];
        double us = 0, usg = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(ctl, 0, 4096, s); hipMemsetAsync(flags, 0, 1024 * 64, s); hipMemsetAsync(pay, 0, (size_t)N * 4096, s);
            hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipLaunchKernelGGL((passes<true, false>), dim3(N), dim3(64), 0, s, pay, flags, ctl, N, S, K, 0);
            hipStreamSynchronize(s);
            us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        hipMemcpy(h, ctl, 16, hipMemcpyDeviceToHost);
        double us2 = 0; u32 h2[4];
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(ctl, 0, 4096, s); hipMemsetAsync(flags, 0, 1024 * 64, s); hipMemsetAsync(pay, 0, (size_t)N * 4096, s);
            hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipLaunchKernelGGL((passes<true, true>), dim3(N), dim3(64), 0, s, pay, flags, ctl, N, S, K, 0);
            hipStreamSynchronize(s);
            us2 = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        hipMemcpy(h2, ctl, 16, hipMemcpyDeviceToHost);
        printf("B %3d workgroups: speculative flag+payload reads %.2f us per pass (timeout %u, bad words %u)\n", N, us2 / K, h2[0], h2[1]);
        // the same passes as one kernel each, captured
        hipGraph_t g; hipGraphExec_t ge;
        hipMemsetAsync(ctl, 0, 4096, s); hipMemsetAsync(pay, 0, (size_t)N * 4096, s);
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL((passes<false, false>), dim3(N), dim3(64), 0, s, pay, flags, ctl, N, S, K, k);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        u32 hb[4]; hipMemcpy(hb, ctl, 16, hipMemcpyDeviceToHost);
        for (int rep = 0; rep < 2; ++rep) {
            hipMemsetAsync(pay, 0, (size_t)N * 4096, s); hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            usg = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        printf("B %3d workgroups, %d passes: neighbour flags %.2f us per pass (timeout %u, bad words %u); kernel per pass in a hipGraph %.2f us (bad words %u)\n",
               N, K, us / K, h[0], h[1], usg / K, hb[1]);
    }
    return 0;
}
