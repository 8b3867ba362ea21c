// Two-sided line sweep with the two halves of a line in SEPARATE waves (same recurrences and the same
// two-sided factor cache as k_line_sweep_tw in smooth.hpp; reference emg3d/core.py:477-1316 line solves).
//
// k_line_sweep_tw keeps both halves of four lines in one wave, so every step carries the arithmetic and the
// selects of both formulations (the left half couples through A_i z_{i-1}, the right half through
// A_{i+1}^T z_{i+1}).  Here wave 2p of a workgroup runs the left halves and wave 2p+1 the right halves of
// the SAME eight lines: each wave executes one specialised instruction stream (the branch on H is
// wave-uniform), a row-load covers eight lines (128 contiguous bytes on the parity-split copies), and the
// halves meet once, at the middle block, through LDS and two workgroup barriers.
#pragma once
#include "smooth.hpp"

static_assert(EMG_RP_BLOCK % 128 == 0, "k_line_sweep_th pairs the waves of a workgroup: whole pairs only");

template <class T, int STAGES, int LPW>
__global__ __launch_bounds__(EMG_RP_BLOCK) void k_line_sweep_th(LineArgs<T> a) {
    typedef unsigned int u32;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int H = __builtin_amdgcn_readfirstlane(wave & 1);     // wave-uniform: 0 = left half, 1 = right half
    const int pair = wave >> 1;
    const int q = lane / LPW;                       // 0..4: rows of my half, >= 5: mirror
    const int g = lane - q * LPW;
    // XCD-aware: workgroup b runs on XCD b % 8 and takes the (b % 8)-th eighth of the line slots, so
    // that lines which share neighbour values (adjacent in Q) meet in the same L2
    EMG_SWEEP_WG(a)
    const i64 gidx = (wg * (blockDim.x >> 7) + pair) * LPW + g;
    i64 jP, jQ;
    if (a.mode == 0) {
        if (gidx >= a.cntA * a.cntB) return;
        const i64 b = gidx / a.cntA, qq = gidx - b * a.cntA;
        jP = 1 + a.cP + 2 * qq;
        jQ = 1 + a.cQ + 2 * b;
    } else {
        if (gidx >= a.cnt) return;
        jQ = a.jQ0 + gidx;
        jP = a.t - 2 * jQ;
    }
    const int L = a.L, P = a.P, Q = a.Q;
    const int nL = (int)a.nC[L];
    const int mid = (int)a.mid;
    const int nLeft = mid, nRight = nL - 1 - mid;   // nRight >= nLeft >= 1
    const int K = H ? nRight : nLeft;       // steps of my half
    const i64 slot = line_slot(a, jP, jQ);
    const i64 nLt = a.nLinesTot;
    const i64 csL = a.cl.st[L], csP = a.cl.st[P], csQ = a.cl.st[Q];
    const double ihP[2] = {a.ih[P][jP - 1], a.ih[P][jP]};
    const double ihQ[2] = {a.ih[Q][jQ - 1], a.ih[Q][jQ]};
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    const FieldLayout& fl = a.fl;
    const i64 jPm = jP - 1, jPp = jP + 1, jQm = jQ - 1, jQp = jQ + 1;
    const i64 nPc = a.nC[P], nPn = a.nC[P] + 1;
    const bool spl = (a.split & 1) != 0;
#define SPC_(v) (spl ? psplit((v), nPc) : (v))
#define SPN_(v) (spl ? psplit((v), nPn) : (v))
#define FL_(vL, vP, vQ) (fl.off[L] + (vL) * fl.st[L][L] + SPN_(vP) * fl.st[L][P] + (vQ) * fl.st[L][Q])
#define FP_(vL, vP, vQ) (fl.off[P] + (vL) * fl.st[P][L] + SPC_(vP) * fl.st[P][P] + (vQ) * fl.st[P][Q])
#define FQ_(vL, vP, vQ) (fl.off[Q] + (vL) * fl.st[Q][L] + SPN_(vP) * fl.st[Q][P] + (vQ) * fl.st[Q][Q])
    const i64 cP0 = SPC_(jP - 1) * csP, cP1 = SPC_(jP) * csP, cq = (jQ - 1) * csQ;

    const bool rowact = q < 5;
    const int rr = rowact ? q : 0;
    const int type = (rr == 0) ? 0 : (rr <= 2 ? 1 : 2);
    const int side = (rr == 0) ? 0 : ((rr - 1) & 1);
    const double sg = side ? -1.0 : 1.0;
    const double tmask = (type == 0) ? 0.0 : 1.0;
    i64 ob[7], os[7];
    i64 fb, sv, suT0;
    double Kc[6];
    double ca = 0.0;
    if (type == 0) {
        ob[0] = FL_(0, jP, jQ);
        ob[1] = FL_(0, jPp, jQ); ob[2] = FL_(0, jPm, jQ); ob[3] = FL_(0, jP, jQp); ob[4] = FL_(0, jP, jQm);
        ob[5] = ob[1]; ob[6] = ob[1];
#pragma unroll
        for (int t = 0; t < 7; ++t) os[t] = fl.st[L][L];
        fb = cP0 + cq; sv = csQ; suT0 = cP1 - cP0;
        Kc[0] = kP[1] * ihP[1]; Kc[1] = kP[0] * ihP[0]; Kc[2] = kQ[1] * ihQ[1]; Kc[3] = kQ[0] * ihQ[0];
        Kc[4] = 0.0; Kc[5] = 0.0;
    } else if (type == 1) {
        const i64 pcell = jPm + side, pnode = side ? jPp : jPm;
        ob[0] = FP_(1, pcell, jQ);
        ob[1] = FL_(1, pnode, jQ); ob[2] = FL_(0, pnode, jQ);
        ob[3] = FQ_(1, pnode, jQ); ob[4] = FQ_(1, pnode, jQm);
        ob[5] = FP_(1, pcell, jQp); ob[6] = FP_(1, pcell, jQm);
        os[0] = fl.st[P][L]; os[1] = fl.st[L][L]; os[2] = fl.st[L][L];
        os[3] = fl.st[Q][L]; os[4] = fl.st[Q][L]; os[5] = fl.st[P][L]; os[6] = fl.st[P][L];
        fb = (side ? cP1 : cP0) + cq; sv = csQ; suT0 = 0;
        const double ihA = ihP[side];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kQ[1] * ihA; Kc[3] = -sg * kQ[0] * ihA;
        Kc[4] = kQ[1] * ihQ[1]; Kc[5] = kQ[0] * ihQ[0];
        ca = sg * 0.5 * ihA;
    } else {
        const i64 qcell = jQm + side, qnode = side ? jQp : jQm;
        ob[0] = FQ_(1, jP, qcell);
        ob[1] = FL_(1, jP, qnode); ob[2] = FL_(0, jP, qnode);
        ob[3] = FP_(1, jP, qnode); ob[4] = FP_(1, jPm, qnode);
        ob[5] = FQ_(1, jPp, qcell); ob[6] = FQ_(1, jPm, qcell);
        os[0] = fl.st[Q][L]; os[1] = fl.st[L][L]; os[2] = fl.st[L][L];
        os[3] = fl.st[P][L]; os[4] = fl.st[P][L]; os[5] = fl.st[Q][L]; os[6] = fl.st[Q][L];
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
#undef SPC_
#undef SPN_
    const bool t0 = (type == 0);

    // byte offsets at block 0 and per-block strides (all < 2^24 resp. 2^32: checked on the host)
    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    char* const eWr = reinterpret_cast<char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    const char* const wB = reinterpret_cast<const char*>(a.fac);
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const char* const hB = reinterpret_cast<const char*>(a.ih[L]);
    u32 wo[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) wo[c] = (u32)(((i64)wpk(rr, c) * nLt + slot) * (i64)sizeof(T));
    const u32 wst = (u32)(15 * nLt * (i64)sizeof(T));
    u32 eo[6], es[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { eo[t] = (u32)(ob[1 + t] * (i64)sizeof(T)); es[t] = (u32)(os[1 + t] * (i64)sizeof(T)); }
    const u32 so = (u32)(ob[0] * (i64)sizeof(T)), ss = (u32)(os[0] * (i64)sizeof(T));
    const u32 zo0 = (u32)(fb * 8), zo1 = (u32)((fb + sv) * 8);
    const u32 zsu = (u32)(suT0 * 8), zsL = (u32)(csL * 8);

    __shared__ T xch[EMG_RP_BLOCK / 64][2][64];
    T* const xu = xch[threadIdx.x >> 6][0];
    T* const xy = xch[threadIdx.x >> 6][1];
    const int sl0 = g;                        // LDS slot of row 0; row c: sl0 + c*LPW
    __shared__ T jn[EMG_RP_BLOCK / 128][2][5 * LPW];     // join at the middle block: [0] z of the right half, [1] x_mid

    // ----------------------------- forward ---------------------------------
    // step k: left block k, right block nL-1-k
    auto fwd_block = [&](int k) -> int { return H ? nL - 1 - k : k; };
    auto load_fwd = [&](int i, TwStep<T>& d) {
        const u32 ic = (u32)(i < 0 ? 0 : (i > nL - 1 ? nL - 1 : i));
        const bool lastb = ((int)ic == nL - 1);
        const u32 su = t0 ? zsu : (lastb ? 0u : zsL);
        const u32 zb = __umul24(ic, zsL);
        d.zf[0] = *reinterpret_cast<const double*>(zB + (zb + zo0));
        d.zf[1] = *reinterpret_cast<const double*>(zB + (zb + zo1));
        d.zf[2] = *reinterpret_cast<const double*>(zB + (zb + zo0 + su));
        d.zf[3] = *reinterpret_cast<const double*>(zB + (zb + zo1 + su));
        d.ihl0 = *reinterpret_cast<const double*>(hB + ic * 8u);
        d.ihl1 = *reinterpret_cast<const double*>(hB + (lastb ? ic : ic + 1u) * 8u);
        const u32 wb = __umul24(ic, wst);
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (wb + wo[c]));
        const u32 ie = ((!t0) && lastb) ? ic - 1u : ic;    // transverse rows of the last block: clamp
        d.S = *reinterpret_cast<const T*>(sB + (so + __umul24(ie, ss)));
#pragma unroll
        for (int t = 0; t < 6; ++t) d.E[t] = *reinterpret_cast<const T*>(eB + (eo[t] + __umul24(ie, es[t])));
    };
    T zprev = Zero<T>::v();
    // right-hand side of block i and the local coupling coefficients
    auto rhs = [&](const TwStep<T>& cur, double& czL, double& czR, double& kL0, double& kL1) -> T {
        kL0 = 0.5 * cur.ihl0; kL1 = 0.5 * cur.ihl1;
        const double rs0 = cur.zf[0] + cur.zf[1], rs1 = cur.zf[2] + cur.zf[3];
        const double cs0 = cur.zf[0] + cur.zf[2], cs1 = cur.zf[1] + cur.zf[3];
        const double g0 = (t0 ? Kc[0] : Kc[0] * kL1) * rs1;
        const double g1 = (t0 ? Kc[1] : Kc[1] * kL0) * rs0;
        T y = cur.S;
        y += g0 * cur.E[0];
        y += g1 * cur.E[1];
        y += (Kc[2] * cs1) * cur.E[2];
        y += (Kc[3] * cs0) * cur.E[3];
        y += (Kc[4] * cs1) * cur.E[4];
        y += (Kc[5] * cs0) * cur.E[5];
        czL = rs0 * cur.ihl0;      // coefficients of A_i     (zeta at L-cell i)
        czR = rs1 * cur.ihl1;      // coefficients of A_{i+1} (zeta at L-cell i+1)
        return y;
    };
    auto fwd_step = [&](int i, const TwStep<T>& cur) {
        const bool lastb = (i == nL - 1);
        const bool full = t0 || !lastb;
        double czL, czR, kL0, kL1;
        T y = rhs(cur, czL, czR, kL0, kL1);
        // left : Y_r = b_r - d_r z_r,   U_r = a_r z_r          (A_i,     zprev = z_{i-1})
        // right: Y_r = b_r - d'_r z_r,  U_0 = z_0, U_r = a'_r  (A_{i+1}, zprev = z_{i+1})
        const double cz = H ? czR : czL;
        const double kk = H ? kL1 : kL0;
        y += ((tmask * kk) * cz) * zprev;
        if (!full) y = Zero<T>::v();
        const double ac = ca * cz;
        T z;
        if (!H) {
            xy[lane] = y;
            xu[lane] = ac * zprev;
            const T Y0 = xy[sl0], Y1 = xy[sl0 + LPW], Y2 = xy[sl0 + 2 * LPW], Y3 = xy[sl0 + 3 * LPW],
                    Y4 = xy[sl0 + 4 * LPW];
            const T su = (xu[sl0 + LPW] + xu[sl0 + 2 * LPW]) + (xu[sl0 + 3 * LPW] + xu[sl0 + 4 * LPW]);
            z = ((cur.W[0] * (Y0 - su) + cur.W[1] * Y1) + (cur.W[2] * Y2 + cur.W[3] * Y3)) + cur.W[4] * Y4;
        } else {
            T uu = zprev;
            if (!t0) { uu = Zero<T>::v(); add_real(uu, ac); }
            xy[lane] = y;
            xu[lane] = uu;
            const T Y0 = xy[sl0], Y1 = xy[sl0 + LPW], Y2 = xy[sl0 + 2 * LPW], Y3 = xy[sl0 + 3 * LPW],
                    Y4 = xy[sl0 + 4 * LPW];
            const T U0 = xu[sl0];
            const double a1 = real_of(xu[sl0 + LPW]), a2 = real_of(xu[sl0 + 2 * LPW]), a3 = real_of(xu[sl0 + 3 * LPW]),
                         a4 = real_of(xu[sl0 + 4 * LPW]);
            z = ((cur.W[0] * Y0 + cur.W[1] * (Y1 - a1 * U0)) + (cur.W[2] * (Y2 - a2 * U0) + cur.W[3] * (Y3 - a3 * U0))) +
                cur.W[4] * (Y4 - a4 * U0);
        }
        if (full && rowact) *reinterpret_cast<T*>(eWr + (so + __umul24((u32)i, ss))) = z;
        zprev = z;
    };
    if (STAGES == 3) {
        // loads run two steps ahead of the arithmetic (block indices are clamped, the
        // one or two extra prefetches past the last step stay inside the line)
        TwStep<T> bufA, bufB, bufC;
        load_fwd(fwd_block(0), bufA);
        load_fwd(fwd_block(1), bufB);
        int k = 0;
        for (; k + 3 <= K; k += 3) {
            load_fwd(fwd_block(k + 2), bufC);
            fwd_step(fwd_block(k), bufA);
            load_fwd(fwd_block(k + 3), bufA);
            fwd_step(fwd_block(k + 1), bufB);
            load_fwd(fwd_block(k + 4), bufB);
            fwd_step(fwd_block(k + 2), bufC);
        }
        if (k < K) fwd_step(fwd_block(k), bufA);
        if (k + 1 < K) fwd_step(fwd_block(k + 1), bufB);
    } else {
        TwStep<T> bufA, bufB;
        load_fwd(fwd_block(0), bufA);
        int k = 0;
        for (; k + 2 <= K - 1; k += 2) {
            load_fwd(fwd_block(k + 1), bufB);
            fwd_step(fwd_block(k), bufA);
            load_fwd(fwd_block(k + 2), bufA);
            fwd_step(fwd_block(k + 1), bufB);
        }
        if (k + 1 <= K - 1) {
            load_fwd(fwd_block(k + 1), bufB);
            fwd_step(fwd_block(k), bufA);
            fwd_step(fwd_block(k + 1), bufB);
        } else {
            fwd_step(fwd_block(k), bufA);
        }
    }

    // ----------------------------- middle ----------------------------------
    if (H && rowact) jn[pair][0][rr * LPW + g] = zprev;          // the right half hands over z_{mid+1}
    __syncthreads();
    if (!H) {
        TwStep<T> cur;
        load_fwd(mid, cur);
        const T zR0 = jn[pair][0][g], zRr = jn[pair][0][rr * LPW + g];
        double czL, czR, kL0, kL1;
        T y = rhs(cur, czL, czR, kL0, kL1);
        y += ((tmask * kL0) * czL) * zprev;                        // - d_r z^L_r
        y -= (ca * czR) * zR0;                                     // - a'_r z^R_0
        y += ((tmask * kL1) * czR) * zRr;                          // - d'_r z^R_r
        xu[lane] = (ca * czL) * zprev;                             // a_r z^L_r
        xy[lane] = y;
        const T Y0 = xy[g], Y1 = xy[LPW + g], Y2 = xy[2 * LPW + g], Y3 = xy[3 * LPW + g], Y4 = xy[4 * LPW + g];
        const T su = (xu[LPW + g] + xu[2 * LPW + g]) + (xu[3 * LPW + g] + xu[4 * LPW + g]);
        const T x = ((cur.W[0] * (Y0 - su) + cur.W[1] * Y1) + (cur.W[2] * Y2 + cur.W[3] * Y3)) + cur.W[4] * Y4;
        if (rowact) {
            *reinterpret_cast<T*>(eWr + (so + __umul24((u32)mid, ss))) = x;
            jn[pair][1][rr * LPW + g] = x;
        }
    }
    __syncthreads();
    zprev = jn[pair][1][rr * LPW + g];                            // both halves continue from x_mid

    // ----------------------------- backward --------------------------------
    // step k: left block mid-1-k, right block mid+1+k
    auto bwd_block = [&](int k) -> int { return H ? mid + 1 + k : mid - 1 - k; };
    auto load_bwd = [&](int i, TwBack<T>& d) {
        const u32 ic = (u32)(i < 0 ? 0 : (i > nL - 1 ? nL - 1 : i));   // prefetches past the ends are clamped
        const bool lastb = ((int)ic == nL - 1);
        const u32 wb = __umul24(ic, wst);
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (wb + wo[c]));
        const u32 ie = ((!t0) && lastb) ? ic - 1u : ic;
        d.zi = *reinterpret_cast<const T*>(eB + (so + __umul24(ie, ss)));
        const u32 ci = H ? ic : ic + 1u;          // left: A_{i+1} (cell i+1); right: A_i (cell i)
        const u32 zb = __umul24(ci, zsL);
        d.p0 = *reinterpret_cast<const double*>(zB + (zb + zo0));
        d.p1 = *reinterpret_cast<const double*>(zB + (zb + zo1));
        d.ihc = *reinterpret_cast<const double*>(hB + ci * 8u);
    };
    auto bwd_step = [&](int i, const TwBack<T>& bc) {
        const bool lastb = (i == nL - 1);
        const bool full = t0 || !lastb;
        const double cz = (bc.p0 + bc.p1) * bc.ihc;
        const double ac = ca * cz;
        const double dc = ((-0.5 * tmask) * bc.ihc) * cz;
        // left : P1_c = d_c x_c (P1_0 = x_0), P2_c = a_c (real)  -> v_c = a_c x_0 + d_c x_c, v_0 = 0
        // right: P1_c = d_c x_c (P1_0 = 0),   P2_c = a_c x_c     -> v_c = d_c x_c,           v_0 = sum a_c x_c
        T w;
        if (!H) {
            T p1 = dc * zprev;
            if (t0) p1 = zprev;
            T p2 = Zero<T>::v();
            add_real(p2, ac);
            xy[lane] = p1;
            xu[lane] = p2;
            const T Q0 = xy[sl0], Q1 = xy[sl0 + LPW], Q2 = xy[sl0 + 2 * LPW], Q3 = xy[sl0 + 3 * LPW],
                    Q4 = xy[sl0 + 4 * LPW];
            const double r1 = real_of(xu[sl0 + LPW]), r2 = real_of(xu[sl0 + 2 * LPW]), r3 = real_of(xu[sl0 + 3 * LPW]),
                         r4 = real_of(xu[sl0 + 4 * LPW]);
            w = (bc.W[1] * (r1 * Q0 + Q1) + bc.W[2] * (r2 * Q0 + Q2)) + (bc.W[3] * (r3 * Q0 + Q3) + bc.W[4] * (r4 * Q0 + Q4));
        } else {
            T p1 = dc * zprev;
            if (t0) p1 = Zero<T>::v();
            xy[lane] = p1;
            xu[lane] = ac * zprev;
            const T Q1 = xy[sl0 + LPW], Q2 = xy[sl0 + 2 * LPW], Q3 = xy[sl0 + 3 * LPW], Q4 = xy[sl0 + 4 * LPW];
            const T v0 = (xu[sl0 + LPW] + xu[sl0 + 2 * LPW]) + (xu[sl0 + 3 * LPW] + xu[sl0 + 4 * LPW]);
            w = ((bc.W[0] * v0 + bc.W[1] * Q1) + (bc.W[2] * Q2 + bc.W[3] * Q3)) + bc.W[4] * Q4;
        }
        const T x = bc.zi - w;
        if (full && rowact) *reinterpret_cast<T*>(eWr + (so + __umul24((u32)i, ss))) = x;
        zprev = x;
    };
    if (STAGES == 3) {
        TwBack<T> bA, bB, bC;
        load_bwd(bwd_block(0), bA);
        load_bwd(bwd_block(1), bB);
        int k = 0;
        for (; k + 3 <= K; k += 3) {
            load_bwd(bwd_block(k + 2), bC);
            bwd_step(bwd_block(k), bA);
            load_bwd(bwd_block(k + 3), bA);
            bwd_step(bwd_block(k + 1), bB);
            load_bwd(bwd_block(k + 4), bB);
            bwd_step(bwd_block(k + 2), bC);
        }
        if (k < K) bwd_step(bwd_block(k), bA);
        if (k + 1 < K) bwd_step(bwd_block(k + 1), bB);
    } else {
        TwBack<T> bA, bB;
        load_bwd(bwd_block(0), bA);
        int k = 0;
        for (; k + 2 <= K - 1; k += 2) {
            load_bwd(bwd_block(k + 1), bB);
            bwd_step(bwd_block(k), bA);
            load_bwd(bwd_block(k + 2), bA);
            bwd_step(bwd_block(k + 1), bB);
        }
        if (k + 1 <= K - 1) {
            load_bwd(bwd_block(k + 1), bB);
            bwd_step(bwd_block(k), bA);
            bwd_step(bwd_block(k + 1), bB);
        } else {
            bwd_step(bwd_block(k), bA);
        }
    }
}
