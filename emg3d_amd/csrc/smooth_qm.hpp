// Quad-per-line chain kernel, TWO-SIDED with MIRRORED right-half blocks (k_line_sweep_qm) and its factorisation
// (k_line_factor_m).  Same line solves as k_line_sweep_q (smooth_q.hpp; reference emg3d/core.py:477-1316), half
// the chain length.
//
// A line holds the unknowns l_0 .. l_{n-1} (edges along the line) and T_0 .. T_{n-2} (the four transverse edges
// at node i+1).  The reference eliminates them in the natural order l_0, T_0, l_1, T_1, ...  A two-sided
// elimination that groups the unknowns of the right half like the left half -- [l_i; T_i], processed downwards --
// loses 3-4 digits on ill-conditioned lines (lines inside a resistive body: every interior node carries a
// discrete gradient that only eta regularises; measured 1e-8 instead of 2e-12 against 80-bit arithmetic,
// tests/tools/conditioning.py).  The MIRROR image of the natural order does not:
//     left  blocks [l_i; T_i],     i = 0 .. m-1,      eliminated upwards   (as the reference does),
//     right blocks [l_j; T_{j-1}], j = n-1 .. m+2,    eliminated downwards (the reference's order on the reversed line),
//     middle       [l_m; T_m; l_{m+1}]                (6 unknowns) last,
// is as accurate as the one-sided order (2e-12 on the same lines).  In the mirrored grouping the right half runs the
// SAME recurrences as the left half on a reversed index with the sign of the l-T coupling flipped (u -> -u), so both
// halves of a line execute one instruction stream in one wave: lanes 0-31 = left halves of eight lines, lanes
// 32-63 = their right halves (lane = 32 H + 4 g + k; quad = one half-line, lane k = transverse row k+1, every
// exchange inside the quad by DPP as in k_line_sweep_q).  The halves meet once, at the middle block, through two
// cross-half shuffles.
//
// Factor layout [block slot][entry 0..14][line]: slot i < m: W of the left block i; slot j > m+1: W of the right
// block j = [l_j; T_{j-1}]; slots m and m+1: the 21 entries of the symmetric 6x6 middle inverse (unknown order
// l_m, T_m[0..3], l_{m+1}; packed lower triangle p = r (r + 1) / 2 + c; p < 15 in slot m, p - 15 in slot m+1).
#pragma once
#include "smooth_q.hpp"

// W = S^{-1} for a symmetric N x N block via non-pivoting LDL^T (the arithmetic of core.solve, core.py:1447-1582).
template <class T, int N>
__device__ __forceinline__ void invert_sym(const T S[N][N], T W[N][N]) {
    T D[N], Dinv[N], Lm[N][N], Nm[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T dj = S[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= (Lm[j][k] * Lm[j][k]) * D[k];
        D[j] = dj;
        const T inv = recip(dj);
        Dinv[j] = inv;
#pragma unroll
        for (int r = j + 1; r < N; ++r) {
            T v = S[r][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= (Lm[r][k] * Lm[j][k]) * D[k];
            Lm[r][j] = v * inv;
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            T t = -Lm[r][c];
#pragma unroll
            for (int k = c + 1; k < r; ++k) t -= Lm[r][k] * Nm[k][c];
            Nm[r][c] = t;
        }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int cc = 0; cc <= r; ++cc) {
            T t = Zero<T>::v();
#pragma unroll
            for (int m = r; m < N; ++m) {
                const T nr = (m == r) ? Dinv[m] : Nm[m][r] * Dinv[m];
                t += (m == cc) ? nr : nr * Nm[m][cc];
            }
            W[r][cc] = t;
            W[cc][r] = t;
        }
}

// middle block of the mirrored two-sided factorisation
// (n - 1) / 2: the halves have equal length for even n, the LEFT half one block more for odd n
__host__ __device__ __forceinline__ i64 qm_mid(i64 nL) { return (nL - 1) / 2; }

template <class T>
__global__ __launch_bounds__(EMG_LINE_BLOCK) void k_line_factor_m(LineArgs<T> a) {
    // all four colours in one launch: blockIdx.y = colour
    const int cP = blockIdx.y & 1, cQ = blockIdx.y >> 1;
    const i64 cntA = a.nA[cP], idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cntA * a.nB2[cQ]) return;
    const i64 b = idx / cntA, q = idx - b * cntA;
    const i64 jP = 1 + cP + 2 * q, jQ = 1 + cQ + 2 * b;
    const i64 n = a.nC[a.L];
    const i64 m = qm_mid(n);
    const i64 slot = line_slot(a, jP, jQ);
    BlockMat<T> bm;
    T W[5][5];
    // ---- left chain: blocks [l_i; T_i], i = 0 .. m-1 (natural order, as k_line_factor) ----
    for (i64 i = 0; i < m; ++i) {
        line_block(a, i, jP, jQ, bm);
        if (i > 0) schur_left(bm.S, bm.al, bm.dl, W, false);
        invert_block(bm.S, W, false);
        store_block(a, i, slot, W);
    }
    T WL[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) WL[r][c] = W[r][c];
    // ---- right chain: blocks [l_j; T_{j-1}], j = n-1 .. m+2 ----
    // M' = [[m_j, +u_j^T], [u_j, M_TT(j-1)]]; the Schur complement of the outer block j+1 is the left formula
    // with u -> -u (coupling of block j+1 to block j: [[0, 0], [-u_j, D_j]]): S -= (-u_j, d_j) W (-u_j, d_j)^T
    T Sr[5][5];
    for (i64 j = n - 1; j > m + 1; --j) {
        line_block(a, j - 1, jP, jQ, bm);               // M_TT(j-1): rows / columns 1..4
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) Sr[r][c] = (r >= 1 && c >= 1) ? bm.S[r][c] : Zero<T>::v();
        line_block(a, j, jP, jQ, bm);                   // m_j, u_j, d_j (zeta at L-cell j)
        Sr[0][0] = bm.S[0][0];
        double un[5];
        un[0] = 0.0;
#pragma unroll
        for (int r = 1; r < 5; ++r) { add_real(Sr[r][0], bm.al[r]); un[r] = -bm.al[r]; }
        if (j < n - 1) schur_left(Sr, un, bm.dl, W, false);
        invert_block(Sr, W, false);
        store_block(a, j, slot, W);
    }
    // ---- middle [l_m; T_m; l_{m+1}] ----
    T S6[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) S6[r][c] = Zero<T>::v();
    line_block(a, m, jP, jQ, bm);
    {
        T S5[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) S5[r][c] = bm.S[r][c];
        if (m > 0) schur_left(S5, bm.al, bm.dl, WL, false);
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) S6[r][c] = S5[r][c];
    }
    line_block(a, m + 1, jP, jQ, bm);
    {
        T S5[5][5];       // the mirror block [l_{m+1}; T_m]: index 0 = l_{m+1}
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) S5[r][c] = (r >= 1 && c >= 1 && r >= c) ? S6[r][c] : Zero<T>::v();
        S5[0][0] = bm.S[0][0];
        double un[5];
        un[0] = 0.0;
#pragma unroll
        for (int r = 1; r < 5; ++r) { add_real(S5[r][0], bm.al[r]); un[r] = -bm.al[r]; }
        if (m + 2 <= n - 1) schur_left(S5, un, bm.dl, W, false);
        S6[5][5] = S5[0][0];
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            S6[5][r] = S5[r][0];
#pragma unroll
            for (int c = 1; c <= r; ++c) S6[r][c] = S5[r][c];
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r + 1; c < 6; ++c) S6[r][c] = S6[c][r];
    T W6[6][6];
    invert_sym<T, 6>(S6, W6);
    {
        T* dst = a.fac + slot;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const int p = r * (r + 1) / 2 + c;
                const i64 blk = p < 15 ? m : m + 1;
                const int ent = p < 15 ? p : p - 15;
                dst[(blk * 15 + ent) * a.nLinesTot] = W6[r][c];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <class T>
struct QmFwd { T W[5]; T W00; T S, S0; T En, Ef; T E[4]; double f0, f1, ihf; };
template <class T>
struct QmBwd { T W[5]; T zk, z0; double f0, f1, ihf; };

template <class T> __device__ __forceinline__ T xhalf(T v);
template <> __device__ __forceinline__ double xhalf<double>(double v) { return __shfl_xor(v, 32, 64); }
template <> __device__ __forceinline__ c128 xhalf<c128>(c128 v) { return mk(__shfl_xor(v.re, 32, 64), __shfl_xor(v.im, 32, 64)); }

// LPW lines per wave (8, 4, 2 or 1): lane = 32 H + 4 g + k, lanes with g >= LPW idle.  STAGES = 2 | 3: register
// prefetch depth (a step consumes the loads issued STAGES - 1 steps earlier).
template <class T, int LPW, int STAGES>
__global__ __launch_bounds__(EMG_Q_BLOCK) void k_line_sweep_qm(LineArgs<T> a) {
    typedef unsigned int u32;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3;
    const int g = (lane >> 2) & 7;
    const int H = lane >> 5;                        // 0: left half (upwards), 1: right half (downwards)
    if (g >= LPW) return;
    EMG_SWEEP_WG(a)
    const i64 gidx = ((wg * blockDim.x + threadIdx.x) >> 6) * LPW + g;
    i64 jP, jQ;
    if (a.mode == 0) {
        if (gidx >= a.cntA * a.cntB) return;
        const i64 b = gidx / a.cntA, q = gidx - b * a.cntA;
        jP = 1 + a.cP + 2 * q;
        jQ = 1 + a.cQ + 2 * b;
    } else {
        if (gidx >= a.cnt) return;
        jQ = a.jQ0 + gidx;
        jP = a.t - 2 * jQ;
    }
    const int L = a.L, P = a.P, Q = a.Q;
    const int n = (int)a.nC[L];
    const int m = (int)a.mid;                       // middle block [l_m; T_m; l_{m+1}]
    const int nleft = m, nright = n - m - 2;        // nleft == nright or nleft == nright + 1
    const int K = nright;                           // uniform steps; the left half may have one extra (outermost) block
    const bool extra = nleft > nright;
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
    const int rr = k + 1;
    const bool tp = k < 2;
    const int side = k & 1;
    const double sg = side ? -1.0 : 1.0;
    i64 tb, eLb, ob[4], ts, os[4];      // own T edge / neighbour's L edge at node resp. cell 0; the four node-plane neighbours
    i64 fb, sv;
    double Kab, Kbe, Kc[4], ca, K0;     // coefficients of the L-edge neighbour above / below the node, of E[0..3]
    if (tp) {
        const i64 pcell = jPm + side, pnode = side ? jPp : jPm;
        tb = FP_(0, pcell, jQ); ts = fl.st[P][L];
        eLb = FL_(0, pnode, jQ);
        ob[0] = FQ_(0, pnode, jQ); ob[1] = FQ_(0, pnode, jQm); ob[2] = FP_(0, pcell, jQp); ob[3] = FP_(0, pcell, jQm);
        os[0] = fl.st[Q][L]; os[1] = fl.st[Q][L]; os[2] = fl.st[P][L]; os[3] = fl.st[P][L];
        fb = (side ? cP1 : cP0) + cq; sv = csQ;
        const double ihA = side ? ihP[1] : ihP[0];
        Kab = sg * ihA; Kbe = -sg * ihA;
        Kc[0] = sg * kQ[1] * ihA; Kc[1] = -sg * kQ[0] * ihA; Kc[2] = kQ[1] * ihQ[1]; Kc[3] = kQ[0] * ihQ[0];
        ca = sg * 0.5 * ihA;
        K0 = side ? kP[1] * ihP[1] : kP[0] * ihP[0];
    } else {
        const i64 qcell = jQm + side, qnode = side ? jQp : jQm;
        tb = FQ_(0, jP, qcell); ts = fl.st[Q][L];
        eLb = FL_(0, jP, qnode);
        ob[0] = FP_(0, jP, qnode); ob[1] = FP_(0, jPm, qnode); ob[2] = FQ_(0, jPp, qcell); ob[3] = FQ_(0, jPm, qcell);
        os[0] = fl.st[P][L]; os[1] = fl.st[P][L]; os[2] = fl.st[Q][L]; os[3] = fl.st[Q][L];
        fb = cP0 + cq + side * csQ; sv = cP1 - cP0;
        const double ihA = side ? ihQ[1] : ihQ[0];
        Kab = sg * ihA; Kbe = -sg * ihA;
        Kc[0] = sg * kP[1] * ihA; Kc[1] = -sg * kP[0] * ihA; Kc[2] = kP[1] * ihP[1]; Kc[3] = kP[0] * ihP[0];
        ca = sg * 0.5 * ihA;
        K0 = side ? kQ[1] * ihQ[1] : kQ[0] * ihQ[0];
    }
    const i64 o0 = FL_(0, jP, jQ);
#undef FL_
#undef FP_
#undef FQ_
#undef SPC_
#undef SPN_
    // "near" = the L-cell of the block's own l (left: below the node, right: above), "far" = the other one
    const double Kn = H ? Kab : Kbe, Kf = H ? Kbe : Kab;
    const double cah = H ? -ca : ca;                 // the mirrored half runs the same recurrences with u -> -u

    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    char* const eW = reinterpret_cast<char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const char* const hB = reinterpret_cast<const char*>(a.ih[L]);
    const char* const wB = reinterpret_cast<const char*>(a.fac);
    const u32 TS = (u32)sizeof(T);
    const u32 wst = (u32)(15 * nLt) * TS;
    u32 wo[5];
    {
        const int cols[5] = {0, rr, 1 + ((k + 1) & 3), 1 + ((k + 2) & 3), 1 + ((k + 3) & 3)};
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int r1 = rr > cols[c] ? rr : cols[c], c1 = rr > cols[c] ? cols[c] : rr;
            wo[c] = (u32)(((i64)(r1 * (r1 + 1) / 2 + c1) * nLt + slot) * (i64)TS);
        }
    }
    const u32 w00 = (u32)(slot * (i64)TS);
    const u32 tss = (u32)ts * TS, sL = (u32)fl.st[L][L] * TS, zsL = (u32)(csL * 8);
    const u32 tb_ = (u32)tb * TS, eLb_ = (u32)eLb * TS, o0_ = (u32)o0 * TS;
    u32 ob_[4], os_[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ob_[t] = (u32)ob[t] * TS; os_[t] = (u32)os[t] * TS; }
    const u32 zo0 = (u32)(fb * 8), zo1 = (u32)((fb + sv) * 8);
    // per-lane signed steps (two's complement u32) towards the middle
    const u32 dW = H ? 0u - wst : wst, dT = H ? 0u - tss : tss, dL = H ? 0u - sL : sL, dZ = H ? 0u - zsL : zsL,
              dH = H ? 0u - 8u : 8u;
    u32 dE[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dE[t] = H ? 0u - os_[t] : os_[t];

    // offsets of block `li` (its own l index) of my half; the T node is li + 1 (left) / li (right)
    auto set_cursors = [&](int li, u32& cw, u32& ct, u32& cl, u32& cEn, u32& cz, u32& ch, u32 ce[4]) {
        const u32 nd = (u32)(H ? li : li + 1);
        const u32 lf = (u32)(H ? li - 1 : li + 1);
        cw = (u32)li * wst;
        ct = tb_ + nd * tss;
        cl = o0_ + (u32)li * sL;
        cEn = eLb_ + (u32)li * sL;
        cz = lf * zsL;                      // far zeta pair
        ch = lf * 8u;
#pragma unroll
        for (int t = 0; t < 4; ++t) ce[t] = ob_[t] + nd * os_[t];
    };

    // ------------------------------------------------ forward ---------------------------------------------------
    u32 cw, ct, cl, cEn, cz, ch, ce[4];
    T zprev = Zero<T>::v();
    double zn0, zn1, ihn;                               // near zeta pair / 1/hL of the NEXT block to process
    auto load_fwd = [&](QmFwd<T>& d) {
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (cw + wo[c]));
        d.W00 = *reinterpret_cast<const T*>(wB + (cw + w00));
        d.f0 = *reinterpret_cast<const double*>(zB + (cz + zo0));
        d.f1 = *reinterpret_cast<const double*>(zB + (cz + zo1));
        d.ihf = *reinterpret_cast<const double*>(hB + ch);
        d.S = *reinterpret_cast<const T*>(sB + ct);
        d.S0 = *reinterpret_cast<const T*>(sB + cl);
        d.En = *reinterpret_cast<const T*>(eB + cEn);
        d.Ef = *reinterpret_cast<const T*>(eB + (cEn + dL));
#pragma unroll
        for (int t = 0; t < 4; ++t) d.E[t] = *reinterpret_cast<const T*>(eB + ce[t]);
        cw += dW; ct += dT; cl += dL; cEn += dL; cz += dZ; ch += dH;
#pragma unroll
        for (int t = 0; t < 4; ++t) ce[t] += dE[t];
    };
    u32 st_t, st_l;                                      // store cursors
    auto fwd_step = [&](const QmFwd<T>& cur) {
        const double kLn = 0.5 * ihn, kLf = 0.5 * cur.ihf;
        const double rsn = zn0 + zn1, rsf = cur.f0 + cur.f1;
        const double cs0 = zn0 + cur.f0, cs1 = zn1 + cur.f1;
        T y = cur.S;
        cmac(y, cur.En, (Kn * kLn) * rsn);
        cmac(y, cur.Ef, (Kf * kLf) * rsf);
        cmac(y, cur.E[0], Kc[0] * cs1);
        cmac(y, cur.E[1], Kc[1] * cs0);
        cmac(y, cur.E[2], Kc[2] * cs1);
        cmac(y, cur.E[3], Kc[3] * cs0);
        const double cz_ = rsn * ihn;
        cmac(y, zprev, kLn * cz_);                       // -d_k z_k
        T part = cur.En * (K0 * rsn);
        cmsc(part, zprev, cah * cz_);                    // -(+-u_k) z_k
        const T y0 = cur.S0 + quad_add(part);
        const T y1 = quad_rot<1>(y), y2 = quad_rot<2>(y), y3 = quad_rot<3>(y);
        T z = cur.W[0] * y0;
        cmac(z, cur.W[1], y);
        cmac(z, cur.W[2], y1);
        cmac(z, cur.W[3], y2);
        cmac(z, cur.W[4], y3);
        const T z0 = cur.W00 * y0 + quad_add(cur.W[0] * y);
        *reinterpret_cast<T*>(eW + st_t) = z;
        if (k == 0) *reinterpret_cast<T*>(eW + st_l) = z0;
        st_t += dT; st_l += dL;
        zprev = z;
        zn0 = cur.f0; zn1 = cur.f1; ihn = cur.ihf;
    };
    {
        const int li0 = H ? n - 1 : 0;                   // outermost block of my half
        set_cursors(li0, cw, ct, cl, cEn, cz, ch, ce);
        st_t = ct; st_l = cl;
        const u32 cn = (u32)li0 * zsL;
        zn0 = *reinterpret_cast<const double*>(zB + (cn + zo0));
        zn1 = *reinterpret_cast<const double*>(zB + (cn + zo1));
        ihn = *reinterpret_cast<const double*>(hB + (u32)li0 * 8u);
    }
    if (extra && H == 0) {                               // the left half's additional outermost block
        QmFwd<T> b0;
        load_fwd(b0);
        fwd_step(b0);
    }
    if (STAGES == 3) {
        QmFwd<T> bA, bB, bC;
        if (K > 0) load_fwd(bA);
        if (K > 1) load_fwd(bB);
        int s = 0;
        for (; s + 3 <= K - 2; s += 3) {
            load_fwd(bC);
            fwd_step(bA);
            load_fwd(bA);
            fwd_step(bB);
            load_fwd(bB);
            fwd_step(bC);
        }
        if (s < K) {                                     // at most 4 steps left: bA = step s, bB = step s+1
            if (s + 2 < K) load_fwd(bC);
            fwd_step(bA);
            if (s + 1 < K) {
                if (s + 3 < K) load_fwd(bA);
                fwd_step(bB);
                if (s + 2 < K) {
                    fwd_step(bC);
                    if (s + 3 < K) fwd_step(bA);
                }
            }
        }
    } else {
        QmFwd<T> bA, bB;
        if (K > 0) load_fwd(bA);
        int s = 0;
        for (; s + 1 <= K - 1; s += 2) {
            load_fwd(bB);
            fwd_step(bA);
            if (s + 2 < K) load_fwd(bA);
            fwd_step(bB);
        }
        if (s < K) fwd_step(bA);
    }

    // ------------------------------------------------ middle ----------------------------------------------------
    // both quads of a line evaluate the 6 x 6 join redundantly (same loads, same arithmetic); zL / zR = z of the
    // innermost left / right block (0 when that half has no block)
    T X0, xprev;
    {
        const T zo = xhalf<T>(zprev);
        const T zL = H ? zo : zprev, zR = H ? zprev : zo;
        const u32 um = (u32)m;
        const u32 nd = um + 1u;
        const double b0 = *reinterpret_cast<const double*>(zB + (um * zsL + zo0)), b1 = *reinterpret_cast<const double*>(zB + (um * zsL + zo1));
        const double a0 = *reinterpret_cast<const double*>(zB + (nd * zsL + zo0)), a1 = *reinterpret_cast<const double*>(zB + (nd * zsL + zo1));
        const double ihb = *reinterpret_cast<const double*>(hB + um * 8u), iha = *reinterpret_cast<const double*>(hB + nd * 8u);
        const T S = *reinterpret_cast<const T*>(sB + (tb_ + nd * tss));
        const T S0b = *reinterpret_cast<const T*>(sB + (o0_ + um * sL)), S0a = *reinterpret_cast<const T*>(sB + (o0_ + nd * sL));
        const T Eb = *reinterpret_cast<const T*>(eB + (eLb_ + um * sL)), Ea = *reinterpret_cast<const T*>(eB + (eLb_ + nd * sL));
        T E[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) E[t] = *reinterpret_cast<const T*>(eB + (ob_[t] + nd * os_[t]));
        // W_mid: unknown order 0 = l_m, 1..4 = T_m, 5 = l_{m+1}; packed p(r, c) = r (r + 1) / 2 + c
        auto wm = [&](int r, int c) -> T {
            const int r1 = r > c ? r : c, c1 = r > c ? c : r;
            const int p = r1 * (r1 + 1) / 2 + c1;
            const u32 blk = p < 15 ? um : nd;
            const int ent = p < 15 ? p : p - 15;
            return *reinterpret_cast<const T*>(wB + (blk * wst + (u32)(((i64)ent * nLt + slot) * (i64)TS)));
        };
        const T W_kl = wm(rr, 0), W_kk = wm(rr, rr), W_k1 = wm(rr, 1 + ((k + 1) & 3)), W_k2 = wm(rr, 1 + ((k + 2) & 3)),
                W_k3 = wm(rr, 1 + ((k + 3) & 3)), W_kh = wm(rr, 5);
        const T W_ll = wm(0, 0), W_hl = wm(5, 0), W_hh = wm(5, 5);
        const double kLb = 0.5 * ihb, kLa = 0.5 * iha;
        const double rsb = b0 + b1, rsa = a0 + a1, cs0 = b0 + a0, cs1 = b1 + a1;
        T y = S;
        cmac(y, Ea, (Kab * kLa) * rsa);
        cmac(y, Eb, (Kbe * kLb) * rsb);
        cmac(y, E[0], Kc[0] * cs1);
        cmac(y, E[1], Kc[1] * cs0);
        cmac(y, E[2], Kc[2] * cs1);
        cmac(y, E[3], Kc[3] * cs0);
        const double czb = rsb * ihb, cza = rsa * iha;
        cmac(y, zL, kLb * czb);                          // -d_m z^L
        cmac(y, zR, kLa * cza);                          // -d_{m+1} z^R
        T pl = Eb * (K0 * rsb);
        cmsc(pl, zL, ca * czb);                          // l_m:     b - u_m . z^L
        T ph = Ea * (K0 * rsa);
        cmac(ph, zR, ca * cza);                          // l_{m+1}: b + u_{m+1} . z^R
        const T yl = S0b + quad_add(pl), yh = S0a + quad_add(ph);
        const T y1 = quad_rot<1>(y), y2 = quad_rot<2>(y), y3 = quad_rot<3>(y);
        T x = W_kl * yl;
        cmac(x, W_kk, y);
        cmac(x, W_k1, y1);
        cmac(x, W_k2, y2);
        cmac(x, W_k3, y3);
        cmac(x, W_kh, yh);
        const T xl = (W_ll * yl + W_hl * yh) + quad_add(W_kl * y);
        const T xh = (W_hl * yl + W_hh * yh) + quad_add(W_kh * y);
        if (!H) {
            *reinterpret_cast<T*>(eW + (tb_ + nd * tss)) = x;
            if (k == 0) *reinterpret_cast<T*>(eW + (o0_ + um * sL)) = xl;
        } else if (k == 0) {
            *reinterpret_cast<T*>(eW + (o0_ + nd * sL)) = xh;
        }
        xprev = x;
        X0 = H ? xh : xl;
    }

    // ------------------------------------------------ backward --------------------------------------------------
    // outwards from the middle: left blocks m-1 .. 0, right blocks m+2 .. n-1.  The inner neighbour's coefficients
    // (u, d of its own l cell) come from the FAR cell of the block being solved.
    auto load_bwd = [&](QmBwd<T>& d) {
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + (cw + wo[c]));
        d.zk = *reinterpret_cast<const T*>(eB + ct);
        d.z0 = *reinterpret_cast<const T*>(eB + cl);
        d.f0 = *reinterpret_cast<const double*>(zB + (cz + zo0));
        d.f1 = *reinterpret_cast<const double*>(zB + (cz + zo1));
        d.ihf = *reinterpret_cast<const double*>(hB + ch);
        cw -= dW; ct -= dT; cl -= dL; cz -= dZ; ch -= dH;
    };
    auto bwd_step = [&](const QmBwd<T>& bc) {
        const double cz_ = (bc.f0 + bc.f1) * bc.ihf;
        const double ak = cah * cz_;
        const double dk = (-0.5 * bc.ihf) * cz_;
        T v = X0 * ak;
        cmac(v, xprev, dk);
        const T v1 = quad_rot<1>(v), v2 = quad_rot<2>(v), v3 = quad_rot<3>(v);
        T x = bc.zk;
        cmsc(x, bc.W[1], v);
        cmsc(x, bc.W[2], v1);
        cmsc(x, bc.W[3], v2);
        cmsc(x, bc.W[4], v3);
        const T x0 = bc.z0 - quad_add(bc.W[0] * v);
        *reinterpret_cast<T*>(eW + st_t) = x;
        if (k == 0) *reinterpret_cast<T*>(eW + st_l) = x0;
        st_t -= dT; st_l -= dL;
        X0 = x0;
        xprev = x;
    };
    if (K > 0 || extra) {
        // innermost block of my half: left m-1, right m+2 (a half without blocks keeps valid but unused cursors)
        const int lib = H ? (K > 0 ? m + 2 : n - 1) : m - 1;
        u32 dummy;
        set_cursors(lib, cw, ct, cl, dummy, cz, ch, ce);
        st_t = ct; st_l = cl;
        if (STAGES == 3) {
            QmBwd<T> bA, bB, bC;
            if (K > 0) load_bwd(bA);
            if (K > 1) load_bwd(bB);
            int s = 0;
            for (; s + 3 <= K - 2; s += 3) {
                load_bwd(bC);
                bwd_step(bA);
                load_bwd(bA);
                bwd_step(bB);
                load_bwd(bB);
                bwd_step(bC);
            }
            if (s < K) {
                if (s + 2 < K) load_bwd(bC);
                bwd_step(bA);
                if (s + 1 < K) {
                    if (s + 3 < K) load_bwd(bA);
                    bwd_step(bB);
                    if (s + 2 < K) {
                        bwd_step(bC);
                        if (s + 3 < K) bwd_step(bA);
                    }
                }
            }
        } else {
            QmBwd<T> bA, bB;
            if (K > 0) load_bwd(bA);
            int s = 0;
            for (; s + 1 <= K - 1; s += 2) {
                load_bwd(bB);
                bwd_step(bA);
                if (s + 2 < K) load_bwd(bA);
                bwd_step(bB);
            }
            if (s < K) bwd_step(bA);
        }
        if (extra && H == 0) {                           // the left half's outermost block
            QmBwd<T> b0;
            load_bwd(b0);
            bwd_step(b0);
        }
    }
}
