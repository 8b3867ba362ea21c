// Two-sided line sweep on the mirrored factorisation (factor_m.hpp, as k_line_sweep_thm) with the recurrences in AFFINE form
// and everything that does not depend on the chain taken off the chain waves.
//
// A forward step of k_line_sweep_thm is  z_k = W_k (y_k - couplings(z_{k-1}))  -- with the couplings written out
//     z_r = sum_c W[r][c] y_c + sum_{c>=1} (W[r][c] c1_c - W[r][0] c2_c) z'_c  =  a_r + sum_{c=1..4} G[r][c] z'_c      (z' = z_{k-1}),
// and a backward step  x_r = z_r - sum_{c>=1} W[r][c] (ac_c x'_0 + dc_c x'_c)  =  z_r + sum_{c=0..4} Hm[r][c] x'_c     (x' = the inner block's x).
// a, G (forward) and Hm (backward) depend on the model, the factor, the neighbour lines and the source only.  A launch's duration
// on the mid levels of a cycle (one wave per SIMD) IS the instruction stream of its chain wave (~3 ns per instruction, whatever
// the instruction; the two-sided kernel with staged right-hand sides, HISTORY R4.6: 112 + 111 instructions per step, 0.68 us); here the chain wave reads five
// numbers per step from LDS, exchanges its z through LDS and does 4 (forward) / 5 (backward) complex multiply-adds.
//
// Workgroup = one group of 8 lines, 2 + 2 NH waves:
//     waves 0, 1          chain of the left / right half (lane = 8 * row + line as in k_line_sweep_thm; 40 lanes + mirrors of row 0);
//     waves 2 .. 2 NH + 1 helpers: helper j of half H produces the half's steps j, j + NH, ... -- first the forward steps' (a, G),
//                         then the backward steps' Hm -- into a ring of D steps in LDS.
// Hand-over: per helper a counter of the last step it has written (release), per half a counter of the steps the chain wave has
// consumed (a helper does not write step s before step s - D has been read).  The two chain waves meet at the middle of the line
// (the 6 x 6 join of k_line_sweep_thm) through counters as well: a workgroup barrier would include the helpers.
// The forward results stay in LDS (zs: the backward steps read their z there and exchange x through the same slots); only x is
// written to the field.
#pragma once
#include "smooth_thm.hpp"

struct ThaPair { double a, b; };

constexpr int THA_LPW = 8;
constexpr int THA_MAX_DYN_LDS = 140 * 1024;      // >= tha_lds_bytes of 128-block complex lines (135 680 B of ring and z) ...
constexpr int THA_STATIC_LDS = 18 * 1024;        // ... + the static exchange buffers, join and counters (17.9 KB at 8 waves, c128): within the CU's 160 KB
// ring of 8 steps (a power of two: the slot index is a mask); 4 and 12 measured the same, 16 does not fit the CU's LDS at 64-block
// lines (HISTORY R5.14)
template <int NH> constexpr int tha_ring_depth() { return 8; }
// (four helpers per half = 10 waves: the 168-register cap, spills, 2 x slower)
template <class T, int NH>
inline size_t tha_lds_bytes(int nL) {
    const size_t KS = (size_t)((nL + 1) / 2);
    return ((size_t)2 * tha_ring_depth<NH>() * 5 + (size_t)2 * (KS + 2)) * (5 * THA_LPW) * sizeof(T);
}

// (Tried: the chain waves alone on their SIMDs -- 4 NH waves, those that would share SIMD 0 / 1 with the chains leave at once:
// no change with two helpers per half, the 168-register cap with three; profiles/r04_tha_sp_ab.txt.)
template <int NH> constexpr int tha_threads() { return 64 * (2 + 2 * NH); }
// ZS: zeta formed from the width vectors instead of read (level 0 of a model without mu_r: smooth_qc.hpp, smooth_thm.hpp).
// Parity-split working copies (LineArgs::split) and the source-line flags of level 0 (LineArgs::sflag) as in k_line_sweep_thm.
template <class T, int NH, bool ZS = false>
__global__ __launch_bounds__(tha_threads<NH>()) void k_line_sweep_tha(LineArgs<T> a) {
    typedef unsigned int u32;
    constexpr int LPW = THA_LPW, D = tha_ring_depth<NH>(), NW = tha_threads<NH>() / 64, L40 = 5 * LPW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = wave & 1;                         // 0 = left half, 1 = right half
    const int hj = (wave >> 1) - 1;                 // helper index within the half; -1: chain wave
    const int q = lane / LPW;                       // 0..4: rows, >= 5: mirror lanes of row 0
    const int g = lane - q * LPW;
    thm_args_burst(a);
    __shared__ int prod[2][4];                      // [half][helper]: 1 + the last step the helper has written
    __shared__ int cons[2];                         // [half]: steps the chain wave has read
    __shared__ int mid[2];                          // [half]: the chain wave's epoch at the middle join
    if (threadIdx.x < 8) (&prod[0][0])[threadIdx.x] = 0;
    if (threadIdx.x >= 8 && threadIdx.x < 10) cons[threadIdx.x - 8] = 0;
    if (threadIdx.x >= 10 && threadIdx.x < 12) mid[threadIdx.x - 10] = 0;
    __syncthreads();
#ifdef EMG3D_LAB
    const bool ts_on = (a.tile & 256) && blockIdx.x == 0 && lane == 0 && H == 0;
    long long ts[6] = {0, 0, 0, 0, 0, 0};
    __shared__ long long ts0;
#define THA_TS(i) do { if (ts_on) ts[i] = (long long)__builtin_readcyclecounter(); } while (0)
    if (ts_on && hj < 0) ts0 = (long long)__builtin_readcyclecounter();
#else
#define THA_TS(i) do {} while (0)
#endif
    THA_TS(0);
    EMG_SWEEP_WG(a)
    // colour order only (mode 0), and everything through the host-resolved 32-bit copies LineArgs::rs (rp_fits: every array is
    // shorter than 2^32 bytes): no kernel-argument array is indexed by a runtime axis here
    const u32 gidx = (u32)wg * LPW + g;
    const u32 cA = (u32)a.cntA;
    if (gidx >= cA * (u32)a.cntB) return;
    const u32 bq = gidx / cA, qq = gidx - bq * cA;
    const u32 jP = 1u + (u32)a.cP + 2u * qq, jQ = 1u + (u32)a.cQ + 2u * bq;
    const int n = (int)a.rs.nL;
    const int m = (int)a.mid;
    const int K = H ? n - m - 2 : m;                // blocks of my half (host: both halves have at least one)
    const int KS = (n + 1) / 2;
    const u32 slot = a.rs.slot0 + gidx;
    const u32 nLt = (u32)a.nLinesTot;
    const u32 csL = a.rs.csL, csP = a.rs.csP, csQ = a.rs.csQ;
    const double ihP[2] = {a.rs.ihP[jP - 1], a.rs.ihP[jP]};
    const double ihQ[2] = {a.rs.ihQ[jQ - 1], a.rs.ihQ[jQ]};
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    const u32 jPm = jP - 1, jPp = jP + 1, jQm = jQ - 1, jQp = jQ + 1;
    // parity-split copies: the P index of a node / cell array is stored even indices first (psplit, common.hpp)
    const bool spl = (a.split & 1) != 0;
    const u32 nPc = a.rs.nP, nPn = a.rs.nP + 1u;
    // (branch-free: v' = (v >> sh) + (v & sh) * half with sh = 0 / 1 -- the compiler turns the conditional form into divergent
    // control flow through the whole row set-up below)
    const u32 spsh = spl ? 1u : 0u, sphc = spl ? (nPc + 1u) >> 1 : 0u, sphn = spl ? (nPn + 1u) >> 1 : 0u;
    auto spc = [&](u32 v) -> u32 { return (v >> spsh) + (v & spsh) * sphc; };
    auto spn = [&](u32 v) -> u32 { return (v >> spsh) + (v & spsh) * sphn; };
#define FL_(vL, vP, vQ) (a.rs.off[0] + (vL) * a.rs.st[0][0] + spn(vP) * a.rs.st[0][1] + (vQ) * a.rs.st[0][2])
#define FP_(vL, vP, vQ) (a.rs.off[1] + (vL) * a.rs.st[1][0] + spc(vP) * a.rs.st[1][1] + (vQ) * a.rs.st[1][2])
#define FQ_(vL, vP, vQ) (a.rs.off[2] + (vL) * a.rs.st[2][0] + spn(vP) * a.rs.st[2][1] + (vQ) * a.rs.st[2][2])
    const u32 cP0 = spc(jP - 1) * csP, cP1 = spc(jP) * csP, cq = (jQ - 1) * csQ;

    // the row's view of a block: identical to k_line_sweep_thm (smooth_thm.hpp), levels without split copies
    const bool rowact = q < 5;
    const int rr = rowact ? q : 0;
    const int type = (rr == 0) ? 0 : (rr <= 2 ? 1 : 2);
    const int side = (rr == 0) ? 0 : ((rr - 1) & 1);
    const double sg = side ? -1.0 : 1.0;
    const double tmask = (type == 0) ? 0.0 : 1.0;
    u32 ob[7], os[7];
    u32 fb, sv, suT0;
    double Kc[6];
    double ca = 0.0;
    if (type == 0) {
        ob[0] = FL_(0, jP, jQ);
        ob[1] = FL_(0, jPp, jQ); ob[2] = FL_(0, jPm, jQ); ob[3] = FL_(0, jP, jQp); ob[4] = FL_(0, jP, jQm);
        ob[5] = ob[1]; ob[6] = ob[1];
#pragma unroll
        for (int t = 0; t < 7; ++t) os[t] = a.rs.st[0][0];
        fb = cP0 + cq; sv = csQ; suT0 = cP1 - cP0;
        Kc[0] = kP[1] * ihP[1]; Kc[1] = kP[0] * ihP[0]; Kc[2] = kQ[1] * ihQ[1]; Kc[3] = kQ[0] * ihQ[0];
        Kc[4] = 0.0; Kc[5] = 0.0;
    } else if (type == 1) {
        const u32 pcell = jPm + side, pnode = side ? jPp : jPm;
        ob[0] = FP_(1, pcell, jQ);
        ob[1] = FL_(1, pnode, jQ); ob[2] = FL_(0, pnode, jQ);
        ob[3] = FQ_(1, pnode, jQ); ob[4] = FQ_(1, pnode, jQm);
        ob[5] = FP_(1, pcell, jQp); ob[6] = FP_(1, pcell, jQm);
        os[0] = a.rs.st[1][0]; os[1] = a.rs.st[0][0]; os[2] = a.rs.st[0][0];
        os[3] = a.rs.st[2][0]; os[4] = a.rs.st[2][0]; os[5] = a.rs.st[1][0]; os[6] = a.rs.st[1][0];
        fb = (side ? cP1 : cP0) + cq; sv = csQ; suT0 = 0;
        const double ihA = ihP[side];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kQ[1] * ihA; Kc[3] = -sg * kQ[0] * ihA;
        Kc[4] = kQ[1] * ihQ[1]; Kc[5] = kQ[0] * ihQ[0];
        ca = sg * 0.5 * ihA;
    } else {
        const u32 qcell = jQm + side, qnode = side ? jQp : jQm;
        ob[0] = FQ_(1, jP, qcell);
        ob[1] = FL_(1, jP, qnode); ob[2] = FL_(0, jP, qnode);
        ob[3] = FP_(1, jP, qnode); ob[4] = FP_(1, jPm, qnode);
        ob[5] = FQ_(1, jPp, qcell); ob[6] = FQ_(1, jPm, qcell);
        os[0] = a.rs.st[2][0]; os[1] = a.rs.st[0][0]; os[2] = a.rs.st[0][0];
        os[3] = a.rs.st[1][0]; os[4] = a.rs.st[1][0]; os[5] = a.rs.st[2][0]; os[6] = a.rs.st[2][0];
        fb = cP0 + cq + side * csQ; sv = cP1 - cP0; suT0 = 0;
        const double ihA = ihQ[side];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kP[1] * ihA; Kc[3] = -sg * kP[0] * ihA;
        Kc[4] = kP[1] * ihP[1]; Kc[5] = kP[0] * ihP[0];
        ca = sg * 0.5 * ihA;
    }
#undef FL_
#undef FP_
#undef FQ_
    const bool t0 = (type == 0);
    const double cah = H ? -ca : ca;                 // the mirrored half: u -> -u
    // ZS: widths across the line of the four zeta values a step uses (row 0: the 2 x 2 face of one cell; transverse rows: the
    // row's pair at two consecutive cells); zeta = (hx hy) hz: z-lines (hP hQ) hL, x- / y-lines (hP hL) hQ (smooth_thm.hpp)
    double zA[4], zB4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u32 cp = (type == 0) ? jP - 1 + (e >> 1) : (type == 1) ? jP - 1 + side : jP - 1 + (e & 1);
        const u32 cq_ = (type == 0) ? jQ - 1 + (e & 1) : (type == 1) ? jQ - 1 + (e & 1) : jQ - 1 + side;
        zA[e] = ZS ? a.rs.hP[cp] : 0.0;
        zB4[e] = ZS ? a.rs.hQ[cq_] : 0.0;
    }
    const bool zl2 = (a.L == 2);
    auto zeta_of = [&](int e, double hl) -> double {     // (the empty asm keeps the rounded product apart from the additions it feeds)
        double v = zl2 ? (zA[e] * zB4[e]) * hl : (zA[e] * hl) * zB4[e];
        asm volatile("" : "+v"(v));
        return v;
    };
    const char* const wLB = reinterpret_cast<const char*>(a.rs.hL);
    // all lines of the workgroup source-free (LineArgs::sflag, level 0): no source loads
    const bool nosrc = a.sflag != nullptr &&
                       __builtin_amdgcn_ballot_w64(a.sflag[(i64)bsys_ * a.nLinesTot + slot] == 0) == __builtin_amdgcn_ballot_w64(true);

    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    char* const eWr = reinterpret_cast<char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    const char* const wB = reinterpret_cast<const char*>(a.fac);
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const char* const hB = reinterpret_cast<const char*>(a.rs.ihL);
    const u32 TS = (u32)sizeof(T);
    u32 wo[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) wo[c] = ((u32)wpk(rr, c) * nLt + slot) * TS;
    const u32 wst = 15u * nLt * TS;
    u32 eo[6], es[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { eo[t] = ob[1 + t] * TS; es[t] = os[1 + t] * TS; }
    const u32 so = ob[0] * TS, ss = os[0] * TS;
    const u32 zo0 = fb * 8u, zo1 = (fb + sv) * 8u;
    const u32 zsu = suT0 * 8u, zsL = csL * 8u;

    // LDS: ring[half][D][5][40], zs[half][KS + 2][40] (dynamic); exchange buffers of the helper waves and of the middle join
    extern __shared__ __attribute__((aligned(16))) char tha_dyn_lds[];
    const int l40 = rr * LPW + g;
    T* const ring = reinterpret_cast<T*>(tha_dyn_lds) + (size_t)H * (D * 5 * L40) + l40;
    T* const zs = reinterpret_cast<T*>(tha_dyn_lds) + (size_t)2 * (D * 5 * L40) + (size_t)H * ((KS + 2) * L40);
    __shared__ T xy_[NW][64];
    __shared__ ThaPair xc_[NW][64];
    __shared__ T jn[2][6 * LPW];                    // middle join: [0] y (6 rows), [1] x (6 rows)
    T* const xy = xy_[wave];
    ThaPair* const xc = xc_[wave];

    auto own_idx = [&](int ic) -> u32 {             // the row's own index for block ic: row 0 by its L-cell, transverse rows by node - 1
        int v = t0 ? ic : (H ? ic - 1 : ic);
        const int hi = t0 ? n - 1 : n - 2;
        v = v < 0 ? 0 : (v > hi ? hi : v);
        return (u32)v;
    };
    auto fwd_block = [&](int k) -> int { return H ? n - 1 - k : k; };
    auto bwd_block = [&](int k) -> int { return H ? m + 2 + k : m - 1 - k; };
    auto rhs = [&](const TmStep<T>& cur, double& czb, double& cza, double& kLb, double& kLa) -> T {
        kLb = 0.5 * cur.ihl0; kLa = 0.5 * cur.ihl1;
        const double f0 = ZS ? zeta_of(0, cur.zf[0]) : cur.zf[0], f1 = ZS ? zeta_of(1, cur.zf[0]) : cur.zf[1];
        const double f2 = ZS ? zeta_of(2, cur.zf[2]) : cur.zf[2], f3 = ZS ? zeta_of(3, cur.zf[2]) : cur.zf[3];
        const double rs0 = f0 + f1, rs1 = f2 + f3;
        const double cs0 = f0 + f2, cs1 = f1 + f3;
        const double g0 = (t0 ? Kc[0] : Kc[0] * kLa) * rs1;
        const double g1 = (t0 ? Kc[1] : Kc[1] * kLb) * rs0;
        T y = cur.S;
        y += g0 * cur.E[0];
        y += g1 * cur.E[1];
        y += (Kc[2] * cs1) * cur.E[2];
        y += (Kc[3] * cs0) * cur.E[3];
        y += (Kc[4] * cs1) * cur.E[4];
        y += (Kc[5] * cs0) * cur.E[5];
        czb = rs0 * cur.ihl0;
        cza = rs1 * cur.ihl1;
        return y;
    };
    // everything of block-row ix a right-hand side needs (W: factor row of block icc; skipped where the caller loads its own)
    auto load_rhs = [&](u32 ix, TmStep<T>& d) {
        const u32 su = t0 ? zsu : zsL;
        const u32 zb = __umul24(ix, zsL);
        if (ZS) {
            d.zf[0] = *reinterpret_cast<const double*>(wLB + ix * 8u);
            d.zf[2] = *reinterpret_cast<const double*>(wLB + (t0 ? ix : ix + 1u) * 8u);
        } else {
            d.zf[0] = *reinterpret_cast<const double*>(zB + (zb + zo0));
            d.zf[1] = *reinterpret_cast<const double*>(zB + (zb + zo1));
            d.zf[2] = *reinterpret_cast<const double*>(zB + (zb + zo0 + su));
            d.zf[3] = *reinterpret_cast<const double*>(zB + (zb + zo1 + su));
        }
        d.ihl0 = *reinterpret_cast<const double*>(hB + ix * 8u);
        d.ihl1 = *reinterpret_cast<const double*>(hB + (t0 ? ix : ix + 1u) * 8u);
        d.S = nosrc ? Zero<T>::v() : *reinterpret_cast<const T*>(sB + (so + __umul24(ix, ss)));
#pragma unroll
        for (int t = 0; t < 6; ++t) d.E[t] = *reinterpret_cast<const T*>(eB + (eo[t] + __umul24(ix, es[t])));
    };
#ifdef EMG3D_LAB
    long long ts_wait = 0;
#endif
    auto wait_ge = [&](int* p, int v, int& seen) {
        if (seen >= v) return;
#ifdef EMG3D_LAB
        const long long w0 = ts_on ? (long long)__builtin_readcyclecounter() : 0;
#endif
        while ((seen = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < v) __builtin_amdgcn_s_sleep(1);
#ifdef EMG3D_LAB
        if (ts_on) ts_wait += (long long)__builtin_readcyclecounter() - w0;
#endif
    };

    if (hj >= 0) {
        // ================================ helper ================================
        if (!rowact) return;                 // (the mirror lanes only spare the chain waves their exec masks)
        int cons_seen = 0;
        auto publish = [&](int gs) {
            if (lane == 0) __hip_atomic_store(&prod[H][hj], gs + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // ---- forward steps hj, hj + NH, ...: (a, G1..G4) ----
        auto load_f = [&](int k_, TmStep<T>& d) {
            const int ic_ = fwd_block(k_ < K ? k_ : K - 1);
            load_rhs(own_idx(ic_), d);
            const u32 wb = __umul24((u32)ic_, wst);
#pragma unroll
            for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (wb + wo[c]));
        };
        auto produce_f = [&](const TmStep<T>& cur, int k_) {
            double czb, cza, kLb, kLa;
            const T y = rhs(cur, czb, cza, kLb, kLa);
            const double cz = H ? cza : czb;             // the block's own l sits below the node in the left half, above it in the right
            const double kk = H ? kLa : kLb;
            ThaPair cc; cc.a = (tmask * kk) * cz; cc.b = cah * cz;
            xy[lane] = y;
            xc[lane] = cc;
            const T Y0 = xy[g], Y1 = xy[g + LPW], Y2 = xy[g + 2 * LPW], Y3 = xy[g + 3 * LPW], Y4 = xy[g + 4 * LPW];
            const ThaPair C1 = xc[g + LPW], C2 = xc[g + 2 * LPW], C3 = xc[g + 3 * LPW], C4 = xc[g + 4 * LPW];
            const T av = ((cur.W[0] * Y0 + cur.W[1] * Y1) + (cur.W[2] * Y2 + cur.W[3] * Y3)) + cur.W[4] * Y4;
            const T G1 = cur.W[1] * C1.a - cur.W[0] * C1.b, G2 = cur.W[2] * C2.a - cur.W[0] * C2.b;
            const T G3 = cur.W[3] * C3.a - cur.W[0] * C3.b, G4 = cur.W[4] * C4.a - cur.W[0] * C4.b;
            wait_ge(&cons[H], k_ - D + 1, cons_seen);
            if (rowact) {
                T* const s_ = ring + (size_t)((u32)k_ % (u32)D) * (5 * L40);
                s_[0] = av; s_[L40] = G1; s_[2 * L40] = G2; s_[3 * L40] = G3; s_[4 * L40] = G4;
            }
            publish(k_);
        };
        THA_TS(1);
        {
            TmStep<T> bA, bB;
            int k = hj;
            if (k < K) load_f(k, bA);
            for (; k < K; k += 2 * NH) {
                if (k + NH < K) load_f(k + NH, bB);
                produce_f(bA, k);
                if (k + NH < K) {
                    if (k + 2 * NH < K) load_f(k + 2 * NH, bA);
                    produce_f(bB, k + NH);
                }
            }
        }
        THA_TS(2);
        // ---- backward steps hj, hj + NH, ...: Hm0..Hm4 (ring steps K + kb) ----
        struct BwdIn { T W[5]; double p0, p1, ihc; };
        auto load_b = [&](int kb_, BwdIn& d) {
            const int ic_ = bwd_block(kb_);
            const u32 wb = __umul24((u32)ic_, wst);
#pragma unroll
            for (int c = 1; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (wb + wo[c]));
            const int ci = H ? ic_ - 1 : ic_ + 1;        // the inner neighbour's l cell
            const u32 zb = __umul24((u32)ci, zsL);
            if (ZS) {
                d.p0 = *reinterpret_cast<const double*>(wLB + (u32)ci * 8u);       // hL[ci]: the pair is formed in produce_b
            } else {
                d.p0 = *reinterpret_cast<const double*>(zB + (zb + zo0));
                d.p1 = *reinterpret_cast<const double*>(zB + (zb + zo1));
            }
            d.ihc = *reinterpret_cast<const double*>(hB + (u32)ci * 8u);
        };
        auto produce_b = [&](const BwdIn& bc, int kb_) {
            const double cz = (ZS ? zeta_of(0, bc.p0) + zeta_of(1, bc.p0) : bc.p0 + bc.p1) * bc.ihc;
            ThaPair cc; cc.a = cah * cz; cc.b = ((-0.5 * tmask) * bc.ihc) * cz;
            xc[lane] = cc;
            const ThaPair C1 = xc[g + LPW], C2 = xc[g + 2 * LPW], C3 = xc[g + 3 * LPW], C4 = xc[g + 4 * LPW];
            const T h0 = (bc.W[1] * C1.a + bc.W[2] * C2.a) + (bc.W[3] * C3.a + bc.W[4] * C4.a);
            const int gs = K + kb_;
            wait_ge(&cons[H], gs - D + 1, cons_seen);
            if (rowact) {
                T* const s_ = ring + (size_t)((u32)gs % (u32)D) * (5 * L40);
                s_[0] = -h0; s_[L40] = -(bc.W[1] * C1.b); s_[2 * L40] = -(bc.W[2] * C2.b);
                s_[3 * L40] = -(bc.W[3] * C3.b); s_[4 * L40] = -(bc.W[4] * C4.b);
            }
            publish(gs);
        };
        {
            BwdIn bA, bB;
            int k = hj;
            if (k < K) load_b(k, bA);
            for (; k < K; k += 2 * NH) {
                if (k + NH < K) load_b(k + NH, bB);
                produce_b(bA, k);
                if (k + NH < K) {
                    if (k + 2 * NH < K) load_b(k + 2 * NH, bA);
                    produce_b(bB, k + NH);
                }
            }
        }
        THA_TS(3);
#ifdef EMG3D_LAB
        if (ts_on) printf("[tha helper %d] entry %lld setup %lld fwd done %lld bwd done %lld (cycles after the chain wave's entry), waiting for ring space %lld\n",
                          hj, ts[0] - ts0, ts[1] - ts0, ts[2] - ts0, ts[3] - ts0, ts_wait);
#endif
        return;
    }

    // ================================ chain ================================
    // the middle's data first (used after the forward steps): my row at the middle -- left wave = block m; right wave: row 0 =
    // cell m + 1, rows 1..4 = node m + 1 -- and my row of the 6 x 6 middle inverse
    const int icm = H ? m + 1 : m;
    TmStep<T> cur;
    load_rhs((u32)(t0 ? icm : m), cur);
    const int ur = H ? 5 : rr;
    T Wm[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int r1 = ur > c ? ur : c, c1 = ur > c ? c : ur;
        const int p = r1 * (r1 + 1) / 2 + c1;
        const u32 blk = p < 15 ? (u32)m : (u32)m + 1u;
        const int ent = p < 15 ? p : p - 15;
        Wm[c] = *reinterpret_cast<const T*>(wB + (__umul24(blk, wst) + ((u32)ent * nLt + slot) * TS));
    }
    int prod_seen[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) prod_seen[j] = 0;
    auto consumed = [&](int gs) {          // the step's LDS reads have returned (their values were used): its slot may be rewritten
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&cons[H], gs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    THA_TS(1);
    // ----------------------------- forward ---------------------------------
    T* const zrow = zs + l40;                        // my row's slot of a step; zs[0] = z_{-1} = 0
    const T* const zcol = zs + g;                    // row c of my line: zcol[c * LPW]
    zrow[0] = Zero<T>::v();
    T zprev = Zero<T>::v();
    auto fwd_step = [&](int k_, int j_) {
        wait_ge(&prod[H][j_], k_ + 1, prod_seen[j_]);
        const T* const s_ = ring + (size_t)((u32)k_ % (u32)D) * (5 * L40);
        const T av = s_[0], G1 = s_[L40], G2 = s_[2 * L40], G3 = s_[3 * L40], G4 = s_[4 * L40];
        const T* const zc = zcol + (size_t)k_ * L40;
        const T Z1 = zc[LPW], Z2 = zc[2 * LPW], Z3 = zc[3 * LPW], Z4 = zc[4 * LPW];
        T z = av;
        T z2 = G2 * Z2;
        cmac(z, G1, Z1);
        cmac(z2, G4, Z4);
        cmac(z, G3, Z3);
        z = z + z2;
        zrow[(size_t)(k_ + 1) * L40] = z;
        zprev = z;
        consumed(k_);
    };
    {
        int k = 0;
        for (; k + NH <= K; k += NH) {
#pragma unroll
            for (int j = 0; j < NH; ++j) fwd_step(k + j, j);
        }
#pragma unroll
        for (int j = 0; j < NH; ++j) if (k + j < K) fwd_step(k + j, j);
    }
    THA_TS(2);
#ifdef EMG3D_LAB
    ts[5] = ts_wait;
#endif
    // ----------------------------- middle ----------------------------------
    // unknowns 0 = l_m (left wave, row 0), 1..4 = T_m (left wave, rows 1..4), 5 = l_{m+1} (right wave, row 0)
    auto pair_sync = [&](int epoch) {
        if (lane == 0) __hip_atomic_store(&mid[H], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&mid[1 - H], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
    };
    pair_sync(1);                                    // both halves have their last z in zs
    {
        double czb, cza, kLb, kLa;
        T y = rhs(cur, czb, cza, kLb, kLa);
        const T zL = H ? Zero<T>::v() : zprev;
        const int KR = n - m - 2;
        const T zR = (reinterpret_cast<T*>(tha_dyn_lds) + (size_t)2 * (D * 5 * L40) + (size_t)((KS + 2) * L40))[(size_t)KR * L40 + l40];
        if (!H) {
            y += ((tmask * kLb) * czb) * zL;
            y += ((tmask * kLa) * cza) * zR;
            xy[lane] = (ca * czb) * zL;                      // u_m,k z^L_k      -> y(l_m)     -= sum
        } else {
            xy[lane] = (ca * cza) * zR;                      // u_{m+1},k z^R_k  -> y(l_{m+1}) += sum
        }
        const T su = (xy[g + LPW] + xy[g + 2 * LPW]) + (xy[g + 3 * LPW] + xy[g + 4 * LPW]);
        if (t0) y = H ? y + su : y - su;
        if (rowact) {
            if (!H) jn[0][rr * LPW + g] = y;
            else if (t0) jn[0][5 * LPW + g] = y;
        }
        pair_sync(2);
        T x = Zero<T>::v();
#pragma unroll
        for (int c = 0; c < 6; ++c) x += Wm[c] * jn[0][c * LPW + g];
        if (rowact && (!H || t0)) jn[1][(H ? 5 : rr) * LPW + g] = x;
        pair_sync(3);
        if (rowact && (!H || t0)) {
            if (!H) *reinterpret_cast<T*>(eWr + (so + __umul24((u32)m, ss))) = x;
            else *reinterpret_cast<T*>(eWr + (so + __umul24((u32)(m + 1), ss))) = x;       // row 0 of the right wave: l_{m+1}
        }
        // both halves continue outwards from the middle: row 0 = the inner block's l, rows 1..4 = T_m
        zprev = jn[1][(t0 ? (H ? 5 : 0) : rr) * LPW + g];
        zrow[(size_t)(K + 1) * L40] = zprev;
    }
    THA_TS(3);
    // ----------------------------- backward --------------------------------
    // step kb: left block m-1-kb, right block m+2+kb = forward step K-1-kb, whose z sits in zs[K-kb]; x replaces it there and
    // is what the next step's exchange reads
    // (the row's own index moves by one per step -- towards the line's ends: no clamps, no multiplication)
    u32 xo = so + __umul24(own_idx(bwd_block(0)), ss);
    const u32 dxo = H ? ss : 0u - ss;
    auto bwd_step = [&](int kb_, int j_) {
        const int gs = K + kb_;
        wait_ge(&prod[H][j_], gs + 1, prod_seen[j_]);
        const T* const s_ = ring + (size_t)((u32)gs % (u32)D) * (5 * L40);
        const T H0 = s_[0], H1 = s_[L40], H2 = s_[2 * L40], H3 = s_[3 * L40], H4 = s_[4 * L40];
        const int kz = K - kb_;
        const T zi = zrow[(size_t)kz * L40];
        const T* const xc_in = zcol + (size_t)(kz + 1) * L40;
        const T X0 = xc_in[0], X1 = xc_in[LPW], X2 = xc_in[2 * LPW], X3 = xc_in[3 * LPW], X4 = xc_in[4 * LPW];
        T x = zi;
        T x2 = H1 * X1;
        cmac(x, H0, X0);
        cmac(x2, H3, X3);
        cmac(x, H2, X2);
        cmac(x2, H4, X4);
        x = x + x2;
        zrow[(size_t)kz * L40] = x;
        *reinterpret_cast<T*>(eWr + xo) = x;       // (mirror lanes: row 0's value again)
        xo += dxo;
        consumed(gs);
    };
    {
        int k = 0;
        for (; k + NH <= K; k += NH) {
#pragma unroll
            for (int j = 0; j < NH; ++j) bwd_step(k + j, j);
        }
#pragma unroll
        for (int j = 0; j < NH; ++j) if (k + j < K) bwd_step(k + j, j);
    }
    THA_TS(4);
#ifdef EMG3D_LAB
    if (ts_on) printf("[tha chain L] K %d: setup %lld forward %lld middle %lld backward %lld (cycles after entry), waiting for helpers %lld (%lld of it in the forward steps)\n", K,
                      ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], ts[4] - ts[0], ts_wait, ts[5]);
#endif
#undef THA_TS
}
