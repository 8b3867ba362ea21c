// Two-sided line sweep with the halves of a line in separate waves, on the MIRRORED two-sided factorisation of
// factor_m.hpp (k_line_factor_m):
//     left  blocks [l_i; T_i],     i = 0 .. m-1,   eliminated upwards   -- wave 2p of a pair,
//     right blocks [l_j; T_{j-1}], j = n-1 .. m+2, eliminated downwards -- wave 2p+1,
//     middle       [l_m; T_m; l_{m+1}]  (6 unknowns), joined through LDS.
// In this grouping the right half runs the SAME recurrences as the left half on a reversed index with the sign of
// the l-T coupling flipped, so both waves execute one instruction stream; and the elimination order is the reference's
// order resp. its mirror image: a sweep agrees with the reference to rounding (4e-12 at 128^3, where round 1's plain
// two-sided grouping -- right-half blocks [l_i; T_i], k_line_sweep_th in the git history -- was off by 1e-8 on
// ill-conditioned lines).
// Lane = LPW * row + line: rows 0..4 of eight lines (40 lanes), row r of consecutive lines in adjacent lanes (one
// 128-byte segment per row on the parity-split copies); the five rows of a block exchange through a wave-private LDS
// buffer once per step; three-deep register prefetch.
#pragma once
#include <type_traits>
#include "factor_m.hpp"

static_assert(EMG_RP_BLOCK % 128 == 0, "k_line_sweep_thm pairs the waves of a workgroup: whole pairs only");

template <class T>
struct TmStep { T W[5]; T E[6]; T S; double zf[4]; double ihl0, ihl1; };
template <class T>
struct TmBack { T W[5]; T zi; double p0, p1, ihc; };

// LIFO (KL > 0): what the backward pass needs again of the LAST KL forward steps of a half -- the 14 distinct entries
// W[r][1..4] of the block inverse and the five z -- stays in LDS (the kernel runs one 256-thread workgroup per CU: 250
// registers; 147 KB of the CU's 160 KB were idle) instead of going through HBM: those steps neither park z in e nor
// read factor and z back.  Per wave and step 19 numbers x LPW lines (2.4 KB at 8 lines): 15 steps of the 63-64 of a
// 128-block line.  Dynamic LDS: thm_lifo_bytes<T, LPW, KL>().
template <class T, int LPW, int KL>
constexpr size_t thm_lifo_bytes() { return (size_t)(EMG_RP_BLOCK / 64) * KL * 19 * LPW * sizeof(T); }

// ZS: zeta formed from the width vectors instead of read (smooth_qc.hpp: level 0 of a model without mu_r, checked bit for
// bit by the handle): zf[0], zf[2] then carry hL at the two cells of the step, zf[1], zf[3] are unused.
// (Round 4's first kernel for the 64-block levels was a variant of this one whose right-hand sides were staged in LDS by helper
// waves -- profiles/HISTORY.md R4.6; k_line_sweep_tha, smooth_tha.hpp, superseded it and the variant was removed.)

// The members of LineArgs the two-sided kernels (k_line_sweep_thm here, k_line_sweep_tha in smooth_tha.hpp) read, loaded in one
// burst (EMG_ARGS_BURST, common.hpp: otherwise every early exit and mode branch of the prologue waits for "its" member)
template <class T>
__device__ __forceinline__ void thm_args_burst(const LineArgs<T>& a) {
    asm volatile("" :: "s"(a.e), "s"(a.s), "s"(a.fac), "s"(a.zeta), "s"(a.rs.ihL), "s"(a.rs.ihP), "s"(a.rs.ihQ), "s"(a.bt.st),
                 "s"(a.bt.mask), "s"(a.bt.n), "s"(a.xcd), "s"(a.cntA), "s"(a.cntB), "s"(a.cP), "s"(a.cQ), "s"(a.mid),
                 "s"(a.rs.nL), "s"(a.rs.csL), "s"(a.rs.csP), "s"(a.rs.csQ), "s"(a.rs.slot0), "s"(a.rs.off[0]), "s"(a.rs.off[1]),
                 "s"(a.rs.off[2]), "s"(a.nLinesTot), "s"(a.split), "s"(a.sflag));
    asm volatile("" :: "s"(a.rs.st[0][0]), "s"(a.rs.st[0][1]), "s"(a.rs.st[0][2]), "s"(a.rs.st[1][0]), "s"(a.rs.st[1][1]),
                 "s"(a.rs.st[1][2]), "s"(a.rs.st[2][0]), "s"(a.rs.st[2][1]), "s"(a.rs.st[2][2]), "s"(a.mode), "s"(a.rs.nP),
                 "s"(a.rs.hL), "s"(a.rs.hP), "s"(a.rs.hQ), "s"(a.L));
}

template <class T, int STAGES, int LPW, int KL = 0, bool ZS = false>
__global__ __launch_bounds__(EMG_RP_BLOCK) void k_line_sweep_thm(LineArgs<T> a) {
    typedef unsigned int u32;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int H = __builtin_amdgcn_readfirstlane(wave & 1);     // wave-uniform: 0 = left half, 1 = right half
    const int pair = wave >> 1;
    const int q = lane / LPW;                       // 0..4: rows, >= 5: mirror lanes (no stores)
    const int g = lane - q * LPW;
    thm_args_burst(a);
    EMG_SWEEP_WG(a)
    // Everything through the host-resolved 32-bit copies LineArgs::rs (the kernel is admitted only where every array is shorter
    // than 2^32 bytes, MG::twist_ok): no kernel-argument array is indexed by a runtime axis, no 64-bit index arithmetic
    const u32 gidx = ((u32)wg * (blockDim.x >> 7) + pair) * LPW + g;
    u32 jP, jQ, slot;
    if (a.mode == 0) {
        const u32 cA = (u32)a.cntA;
        if (gidx >= cA * (u32)a.cntB) return;
        const u32 bq = gidx / cA, qq = gidx - bq * cA;
        jP = 1u + (u32)a.cP + 2u * qq;
        jQ = 1u + (u32)a.cQ + 2u * bq;
        slot = a.rs.slot0 + gidx;                   // the lines of a colour are numbered consecutively
    } else {
        if (gidx >= (u32)a.cnt) return;
        jQ = (u32)a.jQ0 + gidx;
        jP = (u32)a.t - 2u * jQ;
        slot = (u32)line_slot(a, (i64)jP, (i64)jQ);
    }
    const int n = (int)a.rs.nL;
    const int m = (int)a.mid;
    const int K = H ? n - m - 2 : m;                // blocks of my half
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
    // ZS: widths across the line of the four zeta values a step uses (row 0: the 2 x 2 face of one cell; transverse rows:
    // the row's pair at two consecutive cells), zeta = (hx hy) hz: z-lines (hP hQ) hL, x- / y-lines (hP hL) hQ
    double zA[4], zB4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u32 cp = (type == 0) ? jP - 1 + (e >> 1) : (type == 1) ? jP - 1 + side : jP - 1 + (e & 1);
        const u32 cq_ = (type == 0) ? jQ - 1 + (e & 1) : (type == 1) ? jQ - 1 + (e & 1) : jQ - 1 + side;
        zA[e] = ZS ? a.rs.hP[cp] : 0.0;
        zB4[e] = ZS ? a.rs.hQ[cq_] : 0.0;
    }
    const bool zl2 = (a.L == 2);
    // (the empty asm keeps the rounded product apart from the additions it feeds)
    auto zeta_of = [&](int e, double hl) -> double {
        double v = zl2 ? (zA[e] * zB4[e]) * hl : (zA[e] * hl) * zB4[e];
        asm volatile("" : "+v"(v));
        return v;
    };
    const char* const wLB = reinterpret_cast<const char*>(a.rs.hL);

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

    __shared__ T xch[EMG_RP_BLOCK / 64][2][64];
    T* const xu = xch[threadIdx.x >> 6][0];
    T* const xy = xch[threadIdx.x >> 6][1];
    const int sl0 = g;
    __shared__ T jn[EMG_RP_BLOCK / 128][3][6 * LPW];     // middle join: [0] z of the right half, [1] y (6 rows), [2] x (6 rows)
    // LIFO of the wave: [step][item 0..18][line]; items 0..3: W[0][1..4], 4..13: W[k][j] (1 <= j <= k <= 4), 14..18: z_r
    extern __shared__ __attribute__((aligned(16))) char thm_dyn_lds[];
    T* const lifo = reinterpret_cast<T*>(thm_dyn_lds) + (KL > 0 ? (threadIdx.x >> 6) * (KL * 19 * LPW) + g : 0);
    // forward steps K1 .. K-1 (= backward steps KLe-1 .. 0) live in the LIFO; K1 is a multiple of the main loops' unroll
    // (3), so that the loops below split into branch-free phases (a branch in a loop body costs the counted s_waitcnt)
    const int K1 = (KL > 0) ? (K > KL ? ((K - KL + 2) / 3) * 3 : 0) : K;
    const int KLe = K - K1;
    int li_w[4];
#pragma unroll
    for (int c = 1; c <= 4; ++c)
        li_w[c - 1] = (rr == 0) ? c - 1 : (c <= rr ? 4 + (rr - 1) * rr / 2 + c - 1 : 4 + (c - 1) * c / 2 + rr - 1);
    const int li_z = 14 + rr;
    // index of the row's own data for block ic of my half: row 0 by its L-cell, transverse rows by node - 1
    auto own_idx = [&](int ic) -> u32 {
        int v = t0 ? ic : (H ? ic - 1 : ic);
        const int hi = t0 ? n - 1 : n - 2;
        v = v < 0 ? 0 : (v > hi ? hi : v);              // prefetches past the half are clamped (values unused)
        return (u32)v;
    };
    // ----------------------------- forward ---------------------------------
    auto fwd_block = [&](int k) -> int { return H ? n - 1 - k : k; };
    // all lines of the wave source-free (LineArgs::sflag; smooth_qc.hpp): the forward loops run without the source load
    const bool nosrc = a.sflag != nullptr &&
                       __builtin_amdgcn_ballot_w64(a.sflag[(i64)bsys_ * a.nLinesTot + slot] == 0) == __builtin_amdgcn_ballot_w64(true);
    auto load_step = [&](int ic_, TmStep<T>& d, auto nosrc_) {
        const u32 icc = (u32)(ic_ < 0 ? 0 : (ic_ > n - 1 ? n - 1 : ic_));
        const u32 ix = own_idx(ic_);
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
        const u32 wb = __umul24(icc, wst);
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *(reinterpret_cast<const T*>(wB + (wb + wo[c])));
        if constexpr (decltype(nosrc_)::value) d.S = Zero<T>::v();
        else d.S = *(reinterpret_cast<const T*>(sB + (so + __umul24(ix, ss))));
#pragma unroll
        for (int t = 0; t < 6; ++t) d.E[t] = *(reinterpret_cast<const T*>(eB + (eo[t] + __umul24(ix, es[t]))));
    };
    T zprev = Zero<T>::v();
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
        czb = rs0 * cur.ihl0;      // coupling coefficients from zeta at the cell below the node
        cza = rs1 * cur.ihl1;      // ... above the node
        return y;
    };
    // Offsets of the stores of a step: the row's own index moves by exactly one per EXECUTED step (the clamps of own_idx serve
    // the prefetches past a half's end only), so they run along instead of being rebuilt from the block index
    u32 zst = so + __umul24(own_idx(fwd_block(0)), ss);         // forward: towards the middle
    const u32 dzst = H ? 0u - ss : ss;
    auto fwd_step = [&](int ic_, const TmStep<T>& cur, int k_, auto keep_) {
        double czb, cza, kLb, kLa;
        T y = rhs(cur, czb, cza, kLb, kLa);
        // the block's own l sits below the node in the left half, above it in the right half
        const double cz = H ? cza : czb;
        const double kk = H ? kLa : kLb;
        y += ((tmask * kk) * cz) * zprev;                // - d_k z_k
        xy[lane] = y;
        xu[lane] = (cah * cz) * zprev;                   // (+-u_k) z_k
        const T Y0 = xy[sl0], Y1 = xy[sl0 + LPW], Y2 = xy[sl0 + 2 * LPW], Y3 = xy[sl0 + 3 * LPW], Y4 = xy[sl0 + 4 * LPW];
        const T su = (xu[sl0 + LPW] + xu[sl0 + 2 * LPW]) + (xu[sl0 + 3 * LPW] + xu[sl0 + 4 * LPW]);
        const T z = ((cur.W[0] * (Y0 - su) + cur.W[1] * Y1) + (cur.W[2] * Y2 + cur.W[3] * Y3)) + cur.W[4] * Y4;
        if constexpr (decltype(keep_)::value) {          // kept in LDS: no parking in e
            if (rowact) {
                T* const slot_ = lifo + (k_ - K1) * (19 * LPW);
#pragma unroll
                for (int c = 1; c <= 4; ++c)
                    if (rr == 0 || c <= rr) slot_[li_w[c - 1] * LPW] = cur.W[c];
                slot_[li_z * LPW] = z;
            }
        } else if (rowact) *reinterpret_cast<T*>(eWr + zst) = z;
        zst += dzst;
        zprev = z;
    };
    const std::false_type no_{};
    const std::true_type yes_{};
    const std::integral_constant<bool, (KL > 0)> lifo_{};
    auto forward = [&](auto ns) {
    if (K > 0) {
        if (STAGES == 3) {
            TmStep<T> bufA, bufB, bufC;
            load_step(fwd_block(0), bufA, ns);
            load_step(fwd_block(1), bufB, ns);
            int k = 0;
            for (; k + 3 <= K1; k += 3) {
                load_step(fwd_block(k + 2), bufC, ns);
                fwd_step(fwd_block(k), bufA, k, no_);
                load_step(fwd_block(k + 3), bufA, ns);
                fwd_step(fwd_block(k + 1), bufB, k + 1, no_);
                load_step(fwd_block(k + 4), bufB, ns);
                fwd_step(fwd_block(k + 2), bufC, k + 2, no_);
            }
            if constexpr (KL > 0) {
                for (; k + 3 <= K; k += 3) {
                    load_step(fwd_block(k + 2), bufC, ns);
                    fwd_step(fwd_block(k), bufA, k, yes_);
                    load_step(fwd_block(k + 3), bufA, ns);
                    fwd_step(fwd_block(k + 1), bufB, k + 1, yes_);
                    load_step(fwd_block(k + 4), bufB, ns);
                    fwd_step(fwd_block(k + 2), bufC, k + 2, yes_);
                }
            }
            if (k < K) fwd_step(fwd_block(k), bufA, k, lifo_);
            if (k + 1 < K) fwd_step(fwd_block(k + 1), bufB, k + 1, lifo_);
        } else {
            TmStep<T> bufA, bufB;
            load_step(fwd_block(0), bufA, ns);
            int k = 0;
            auto step2 = [&](int ic_, const TmStep<T>& cur, int k_) {
                if (KL > 0 && k_ >= K1) fwd_step(ic_, cur, k_, yes_); else fwd_step(ic_, cur, k_, no_);
            };
            for (; k + 2 <= K - 1; k += 2) {
                load_step(fwd_block(k + 1), bufB, ns);
                step2(fwd_block(k), bufA, k);
                load_step(fwd_block(k + 2), bufA, ns);
                step2(fwd_block(k + 1), bufB, k + 1);
            }
            if (k + 1 <= K - 1) {
                load_step(fwd_block(k + 1), bufB, ns);
                step2(fwd_block(k), bufA, k);
                step2(fwd_block(k + 1), bufB, k + 1);
            } else {
                step2(fwd_block(k), bufA, k);
            }
        }
    }
    };
    if (nosrc) forward(std::true_type{}); else forward(std::false_type{});

    // ----------------------------- middle ----------------------------------
    // unknowns 0 = l_m (left wave, row 0), 1..4 = T_m (left wave, rows 1..4), 5 = l_{m+1} (right wave, row 0).
    // Both waves load node m+1's data; the right wave's row-0 lanes work on cell m+1.
    if (H && rowact) jn[pair][0][rr * LPW + g] = zprev;             // z^R (rows 1..4 are used)
    __syncthreads();
    {
        // my row's data at the middle: left wave = block m as usual; right wave: row 0 = cell m+1, rows 1..4 = node m+1
        const int icm = H ? m + 1 : m;
        TmStep<T> cur;
        {
            const u32 ix = (u32)(t0 ? icm : m);
            const u32 su = t0 ? zsu : zsL;
            const u32 zb = __umul24(ix, zsL);
            if (ZS) {
                cur.zf[0] = *reinterpret_cast<const double*>(wLB + ix * 8u);
                cur.zf[2] = *reinterpret_cast<const double*>(wLB + (t0 ? ix : ix + 1u) * 8u);
            } else {
                cur.zf[0] = *reinterpret_cast<const double*>(zB + (zb + zo0));
                cur.zf[1] = *reinterpret_cast<const double*>(zB + (zb + zo1));
                cur.zf[2] = *reinterpret_cast<const double*>(zB + (zb + zo0 + su));
                cur.zf[3] = *reinterpret_cast<const double*>(zB + (zb + zo1 + su));
            }
            cur.ihl0 = *reinterpret_cast<const double*>(hB + ix * 8u);
            cur.ihl1 = *reinterpret_cast<const double*>(hB + (t0 ? ix : ix + 1u) * 8u);
            cur.S = nosrc ? Zero<T>::v() : *reinterpret_cast<const T*>(sB + (so + __umul24(ix, ss)));
#pragma unroll
            for (int t = 0; t < 6; ++t) cur.E[t] = *reinterpret_cast<const T*>(eB + (eo[t] + __umul24(ix, es[t])));
        }
        // my row of the 6 x 6 middle inverse: unknown index of my row
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
        double czb, cza, kLb, kLa;
        T y = rhs(cur, czb, cza, kLb, kLa);
        const T zL = H ? Zero<T>::v() : zprev;
        const T zR = jn[pair][0][rr * LPW + g];
        // transverse rows (left wave): - d_m z^L - d_{m+1} z^R; row sums for the two l rows through xu
        if (!H) {
            y += ((tmask * kLb) * czb) * zL;
            y += ((tmask * kLa) * cza) * zR;
            xu[lane] = (ca * czb) * zL;                      // u_m,k z^L_k      -> y(l_m)     -= sum
        } else {
            xu[lane] = (ca * cza) * zR;                      // u_{m+1},k z^R_k  -> y(l_{m+1}) += sum
        }
        const T su = (xu[sl0 + LPW] + xu[sl0 + 2 * LPW]) + (xu[sl0 + 3 * LPW] + xu[sl0 + 4 * LPW]);
        if (t0) y = H ? y + su : y - su;
        if (rowact) {
            if (!H) jn[pair][1][rr * LPW + g] = y;
            else if (t0) jn[pair][1][5 * LPW + g] = y;
        }
        __syncthreads();
        T x = Zero<T>::v();
#pragma unroll
        for (int c = 0; c < 6; ++c) x += Wm[c] * jn[pair][1][c * LPW + g];
        if (rowact && (!H || t0)) {
            if (!H) *reinterpret_cast<T*>(eWr + (so + __umul24((u32)m, ss))) = x;
            else *reinterpret_cast<T*>(eWr + (so + __umul24((u32)(m + 1), ss))) = x;       // row 0 of the right wave: l_{m+1}
            jn[pair][2][(H ? 5 : rr) * LPW + g] = x;
        }
        __syncthreads();
        // both halves continue outwards from the middle: row 0 = the inner block's l, rows 1..4 = T_m
        zprev = jn[pair][2][(t0 ? (H ? 5 : 0) : rr) * LPW + g];
    }

    // ----------------------------- backward --------------------------------
    // step k: left block m-1-k, right block m+2+k; the inner neighbour's l cell: left ic+1, right ic-1
    auto bwd_block = [&](int k) -> int { return H ? m + 2 + k : m - 1 - k; };
    // src_: 1 = the step is in the LIFO, 2 = it is not, 0 = decide here (wave-uniform branch: transition iterations only)
    auto load_bwd = [&](int ic_, TmBack<T>& d, int kb_, auto src_) {
        constexpr int SRC = decltype(src_)::value;
        const u32 icc = (u32)(ic_ < 0 ? 0 : (ic_ > n - 1 ? n - 1 : ic_));
        if (KL > 0 && (SRC == 1 || (SRC == 0 && kb_ < KLe))) {      // the step's factor rows and z are in the LIFO
            const T* const slot_ = lifo + (KLe - 1 - kb_) * (19 * LPW);
#pragma unroll
            for (int c = 1; c <= 4; ++c) d.W[c] = slot_[li_w[c - 1] * LPW];
            d.zi = slot_[li_z * LPW];
        } else {
            const u32 wb = __umul24(icc, wst);
#pragma unroll
            for (int c = 1; c < 5; ++c) d.W[c] = *(reinterpret_cast<const T*>(wB + (wb + wo[c])));
            d.zi = *(reinterpret_cast<const T*>(eB + (so + __umul24(own_idx(ic_), ss))));
        }
        int ci = H ? (int)icc - 1 : (int)icc + 1;
        ci = ci < 0 ? 0 : (ci > n - 1 ? n - 1 : ci);
        const u32 zb = __umul24((u32)ci, zsL);
        if (ZS) {
            d.p0 = *reinterpret_cast<const double*>(wLB + (u32)ci * 8u);      // hL[ci]: the pair is formed in bwd_step
        } else {
            d.p0 = *reinterpret_cast<const double*>(zB + (zb + zo0));
            d.p1 = *reinterpret_cast<const double*>(zB + (zb + zo1));
        }
        d.ihc = *reinterpret_cast<const double*>(hB + (u32)ci * 8u);
    };
    u32 xst = so + __umul24(own_idx(bwd_block(0)), ss);         // backward: away from the middle
    const u32 dxst = H ? ss : 0u - ss;
    auto bwd_step = [&](int ic_, const TmBack<T>& bc) {
        const double cz = (ZS ? zeta_of(0, bc.p0) + zeta_of(1, bc.p0) : bc.p0 + bc.p1) * bc.ihc;
        const double ac = cah * cz;
        const double dc = ((-0.5 * tmask) * bc.ihc) * cz;
        // P1_c = d_c x_c (P1_0 = x_0 of the inner block), P2_c = (+-)a_c  ->  v_c = a_c x_0 + d_c x_c, v_0 = 0
        T p1 = dc * zprev;
        if (t0) p1 = zprev;
        T p2 = Zero<T>::v();
        add_real(p2, ac);
        xy[lane] = p1;
        xu[lane] = p2;
        const T Q0 = xy[sl0], Q1 = xy[sl0 + LPW], Q2 = xy[sl0 + 2 * LPW], Q3 = xy[sl0 + 3 * LPW], Q4 = xy[sl0 + 4 * LPW];
        const double r1 = real_of(xu[sl0 + LPW]), r2 = real_of(xu[sl0 + 2 * LPW]), r3 = real_of(xu[sl0 + 3 * LPW]),
                     r4 = real_of(xu[sl0 + 4 * LPW]);
        const T w = (bc.W[1] * (r1 * Q0 + Q1) + bc.W[2] * (r2 * Q0 + Q2)) + (bc.W[3] * (r3 * Q0 + Q3) + bc.W[4] * (r4 * Q0 + Q4));
        const T x = bc.zi - w;
        if (rowact) *reinterpret_cast<T*>(eWr + xst) = x;
        xst += dxst;
        zprev = x;
    };
    const std::integral_constant<int, 0> any_{};
    const std::integral_constant<int, 1> lds_{};
    const std::integral_constant<int, 2> mem_{};
    if (K > 0) {
        if (STAGES == 3) {
            TmBack<T> bA, bB, bC;
            load_bwd(bwd_block(0), bA, 0, any_);
            load_bwd(bwd_block(1), bB, 1, any_);
            int k = 0;
            if constexpr (KL > 0) {
                for (; k + 3 <= K && k + 4 < KLe; k += 3) {          // every load of the iteration from the LIFO
                    load_bwd(bwd_block(k + 2), bC, k + 2, lds_);
                    bwd_step(bwd_block(k), bA);
                    load_bwd(bwd_block(k + 3), bA, k + 3, lds_);
                    bwd_step(bwd_block(k + 1), bB);
                    load_bwd(bwd_block(k + 4), bB, k + 4, lds_);
                    bwd_step(bwd_block(k + 2), bC);
                }
                for (; k + 3 <= K && k + 2 < KLe; k += 3) {          // transition
                    load_bwd(bwd_block(k + 2), bC, k + 2, any_);
                    bwd_step(bwd_block(k), bA);
                    load_bwd(bwd_block(k + 3), bA, k + 3, any_);
                    bwd_step(bwd_block(k + 1), bB);
                    load_bwd(bwd_block(k + 4), bB, k + 4, any_);
                    bwd_step(bwd_block(k + 2), bC);
                }
            }
            for (; k + 3 <= K; k += 3) {
                load_bwd(bwd_block(k + 2), bC, k + 2, mem_);
                bwd_step(bwd_block(k), bA);
                load_bwd(bwd_block(k + 3), bA, k + 3, mem_);
                bwd_step(bwd_block(k + 1), bB);
                load_bwd(bwd_block(k + 4), bB, k + 4, mem_);
                bwd_step(bwd_block(k + 2), bC);
            }
            if (k < K) bwd_step(bwd_block(k), bA);
            if (k + 1 < K) bwd_step(bwd_block(k + 1), bB);
        } else {
            TmBack<T> bA, bB;
            load_bwd(bwd_block(0), bA, 0, any_);
            int k = 0;
            for (; k + 2 <= K - 1; k += 2) {
                load_bwd(bwd_block(k + 1), bB, k + 1, any_);
                bwd_step(bwd_block(k), bA);
                load_bwd(bwd_block(k + 2), bA, k + 2, any_);
                bwd_step(bwd_block(k + 1), bB);
            }
            if (k + 1 <= K - 1) {
                load_bwd(bwd_block(k + 1), bB, k + 1, any_);
                bwd_step(bwd_block(k), bA);
                bwd_step(bwd_block(k + 1), bB);
            } else {
                bwd_step(bwd_block(k), bA);
            }
        }
    }
}
