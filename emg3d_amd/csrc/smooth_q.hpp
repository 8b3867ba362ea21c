// Quad-per-line chain kernel: the line smoother sweep with FOUR lanes per line and every exchange inside the
// quad done with DPP (no LDS, no barrier).  Same recurrences and the same cached one-sided block factorisation
// as k_line_sweep_rp (smooth.hpp; reference emg3d/core.py:477-1316 line solves, core.py:1447-1582 band LDL^T in
// its block form, natural order: the elimination order of the reference):
//     forward : z_i = W_i (b_i - A_i z_{i-1})           backward: x_i = z_i - W_i A_{i+1}^T x_{i+1}
//
// Lane k of a quad owns the transverse unknown k+1 of every block (rows 1,2: the two P-directed edges at node
// i+1, rows 3,4: the two Q-directed edges); row 0 (the edge along the line) has no lane of its own: its
// right-hand side is the sum of one term per lane (each lane already holds the neighbour value and the zeta
// pair of its side), and its solution component needs one more quad sum -- which is off the dependent chain,
// because A_i has a zero first column.  Per block step a lane
//   * forms its row of the right-hand side (six neighbour values x coefficients from its zeta pairs),
//   * adds the coupling to the previous block: row k gets -d_k z_k, row 0 gets -sum_k a_k z_k (quad sum),
//   * gathers the other three y values of the quad with three quad rotations and multiplies with its row of
//     the cached symmetric inverse W_i.
// 16 lines per wave, all 64 lanes active (the lane-group kernels use 40 of 64), ~3 x fewer instructions per
// block than k_line_sweep_th and a dependent chain of two DPP stages instead of an LDS round trip.
//
// Why one-sided: the two-sided elimination of k_line_sweep_th/_tw is 10^3-10^4 x less accurate on the
// ill-conditioned lines of the benchmark models (lines inside a resistive body: every interior node of a line
// carries a discrete gradient, a null vector of the curl-curl part that only eta regularises; condition
// ~ 1 / (omega mu sigma h^2) ~ 1e4..1e6).  Measured on the 128^3 model (tests/tools/conditioning.py, error of
// ONE line solve against 80-bit arithmetic): reference order 2e-12, two-sided 1e-8.  This kernel keeps the
// reference's elimination order, so a sweep agrees with the reference to ~1e-12 at every size.
#pragma once
#include "smooth.hpp"

// quad rotation: lane k of every quad reads lane (k + R) % 4.  __builtin_amdgcn_mov_dpp with bound_ctrl needs no
// "old" operand (update_dpp(0, ...) costs a v_mov_b32 0 in front of every DPP move).
template <int R>
__device__ __forceinline__ double quad_rot(double v) {
    constexpr int ctrl = ((0 + R) & 3) | (((1 + R) & 3) << 2) | (((2 + R) & 3) << 4) | (((3 + R) & 3) << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int R>
__device__ __forceinline__ c128 quad_rot(c128 v) { return mk(quad_rot<R>(v.re), quad_rot<R>(v.im)); }
// sum over the quad (butterfly: xor 1, xor 2)
__device__ __forceinline__ double quad_add(double v) {
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0xb1, 0xf, 0xf, true),      // quad_perm [1,0,3,2]
                          __builtin_amdgcn_mov_dpp(__double2loint(v), 0xb1, 0xf, 0xf, true));
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4e, 0xf, 0xf, true),      // quad_perm [2,3,0,1]
                          __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4e, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ c128 quad_add(c128 v) { return mk(quad_add(v.re), quad_add(v.im)); }

template <class T>
struct QFwd {           // what a lane loads for one forward block step
    T W[5];             // its row of W_i: [0] column 0, [1] diagonal, [2..4] the other columns in rotation order
    T W00;
    T S, S0;            // source of the row's own edge / of the edge along the line
    T E[6];
    double n0, n1;      // zeta pair of the row's side at cell i+1
    double ihl1;        // 1 / hL[i+1]
};
template <class T>
struct QBwd {
    T W[5];
    T zk, z0;
    double p0, p1, ihn; // zeta pair at cell i+1, 1 / hL[i+1]
};

#ifndef EMG_Q_BLOCK
#define EMG_Q_BLOCK 256
#endif

// LPW lines per wave (16, 8, 4 or 2: the lanes beyond 4 LPW idle).  A launch with few lines is bound by the
// latency of its loads (a step consumes what was requested two steps earlier), not by lanes: fewer lines per
// wave = more waves = more requests in flight, and the SIMDs they occupy would be idle anyway.
template <class T, int STAGES, int LPW>
__global__ __launch_bounds__(EMG_Q_BLOCK) void k_line_sweep_q(LineArgs<T> a) {
    typedef unsigned int u32;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3;                         // row k + 1
    const int g = lane >> 2;                        // line of the wave
    if (g >= LPW) return;
    EMG_SWEEP_WG(a)
    const i64 gidx = ((wg * blockDim.x + threadIdx.x) >> 6) * LPW + g;
    i64 jP, jQ;
    if (a.mode == 0 && a.tile) {
        // workgroup = one chunk of LPW lines along P x (waves per workgroup) consecutive rows of the colour along Q
        const i64 nPc = (a.cntA + LPW - 1) / LPW;
        const i64 qb = wg / nPc, pc = wg - qb * nPc;
        const i64 q = pc * LPW + g, b = qb * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (q >= a.cntA || b >= a.cntB) return;
        jP = 1 + a.cP + 2 * q;
        jQ = 1 + a.cQ + 2 * b;
    } else if (a.mode == 0) {
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
    const int nL = (int)a.nC[L];
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
    const bool tp = k < 2;                          // rows 1,2: P-directed edges; rows 3,4: Q-directed
    const int side = k & 1;
    const double sg = side ? -1.0 : 1.0;
    i64 ob[7], os[7];
    i64 fb, sv;                                     // zeta pair of the row's side: base, stride inside the pair
    double Kc[6], ca, K0;
    if (tp) {
        const i64 pcell = jPm + side, pnode = side ? jPp : jPm;
        ob[0] = FP_(1, pcell, jQ);
        ob[1] = FL_(1, pnode, jQ); ob[2] = FL_(0, pnode, jQ);
        ob[3] = FQ_(1, pnode, jQ); ob[4] = FQ_(1, pnode, jQm);
        ob[5] = FP_(1, pcell, jQp); ob[6] = FP_(1, pcell, jQm);
        os[0] = fl.st[P][L]; os[1] = fl.st[L][L]; os[2] = fl.st[L][L];
        os[3] = fl.st[Q][L]; os[4] = fl.st[Q][L]; os[5] = fl.st[P][L]; os[6] = fl.st[P][L];
        fb = (side ? cP1 : cP0) + cq; sv = csQ;
        const double ihA = side ? ihP[1] : ihP[0];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kQ[1] * ihA; Kc[3] = -sg * kQ[0] * ihA;
        Kc[4] = kQ[1] * ihQ[1]; Kc[5] = kQ[0] * ihQ[0];
        ca = sg * 0.5 * ihA;
        K0 = side ? kP[1] * ihP[1] : kP[0] * ihP[0];
    } else {
        const i64 qcell = jQm + side, qnode = side ? jQp : jQm;
        ob[0] = FQ_(1, jP, qcell);
        ob[1] = FL_(1, jP, qnode); ob[2] = FL_(0, jP, qnode);
        ob[3] = FP_(1, jP, qnode); ob[4] = FP_(1, jPm, qnode);
        ob[5] = FQ_(1, jPp, qcell); ob[6] = FQ_(1, jPm, qcell);
        os[0] = fl.st[Q][L]; os[1] = fl.st[L][L]; os[2] = fl.st[L][L];
        os[3] = fl.st[P][L]; os[4] = fl.st[P][L]; os[5] = fl.st[Q][L]; os[6] = fl.st[Q][L];
        fb = cP0 + cq + side * csQ; sv = cP1 - cP0;
        const double ihA = side ? ihQ[1] : ihQ[0];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kP[1] * ihA; Kc[3] = -sg * kP[0] * ihA;
        Kc[4] = kP[1] * ihP[1]; Kc[5] = kP[0] * ihP[0];
        ca = sg * 0.5 * ihA;
        K0 = side ? kQ[1] * ihQ[1] : kQ[0] * ihQ[0];
    }
    const i64 o0 = FL_(0, jP, jQ);                  // the edge along the line, block 0
#undef FL_
#undef FP_
#undef FQ_
#undef SPC_
#undef SPN_

    // Addressing as in k_line_sweep_rp: uniform base pointers + 32-bit per-lane BYTE offsets that advance by
    // a per-lane stride per block (the host selects this kernel only when every field array is < 4 GiB); the
    // factor base is a 64-bit uniform pointer (the factor of a direction may exceed 4 GiB).
    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    char* const eW = reinterpret_cast<char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const double* const hB = a.ih[L];
    const i64 wstep = 15 * nLt * (i64)sizeof(T);
    u32 wo[5];                                       // entry offsets of the lane's row inside one block record
    {
        const int cols[5] = {0, rr, 1 + ((k + 1) & 3), 1 + ((k + 2) & 3), 1 + ((k + 3) & 3)};
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            // wpk(rr, col) with a per-lane rr: evaluate the packed index arithmetically (no local array)
            const int r1 = rr > cols[c] ? rr : cols[c], c1 = rr > cols[c] ? cols[c] : rr;
            wo[c] = (u32)(((i64)(r1 * (r1 + 1) / 2 + c1) * nLt + slot) * (i64)sizeof(T));
        }
    }
    const u32 w00 = (u32)(slot * (i64)sizeof(T));
    const u32 ss = (u32)(os[0] * (i64)sizeof(T));               // stride of the row's own edge
    const u32 sL = (u32)(fl.st[L][L] * (i64)sizeof(T));         // stride of the edge along the line
    const u32 so_base = (u32)(ob[0] * (i64)sizeof(T));
    const u32 o0_base = (u32)(o0 * (i64)sizeof(T));
    u32 es[6], eb_[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { eb_[t] = (u32)(ob[1 + t] * (i64)sizeof(T)); es[t] = (u32)(os[1 + t] * (i64)sizeof(T)); }
    const u32 zo0 = (u32)(fb * 8), zo1 = (u32)((fb + sv) * 8), zsL = (u32)(csL * 8);

    // ----------------------------- forward ---------------------------------
    // load cursors (blocks are loaded in ascending order) and store cursors (one step behind)
    const char* wB = reinterpret_cast<const char*>(a.fac);
    u32 l_so = so_base, l_o0 = o0_base, l_z = zsL, l_e[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) l_e[t] = eb_[t];
    const double* l_h = hB + 1;
    auto load_fwd = [&](int i, QFwd<T>& d) {
        const bool lastb = (i == nL - 1);
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + wo[c]);
        d.W00 = *reinterpret_cast<const T*>(wB + w00);
        // cell i+1 (clamped on the last block)
        d.n0 = *reinterpret_cast<const double*>(zB + ((lastb ? l_z - zsL : l_z) + zo0));
        d.n1 = *reinterpret_cast<const double*>(zB + ((lastb ? l_z - zsL : l_z) + zo1));
        d.ihl1 = lastb ? l_h[-1] : l_h[0];
        d.S = *reinterpret_cast<const T*>(sB + l_so);
        d.S0 = *reinterpret_cast<const T*>(sB + l_o0);
        // E[0] is the neighbour's edge along the line at index i+1: it does not exist on the last block (clamped,
        // the row is zeroed there); all other neighbour values sit on node planes and exist for every block
        d.E[0] = *reinterpret_cast<const T*>(eB + (lastb ? l_e[0] - es[0] : l_e[0]));
#pragma unroll
        for (int t = 1; t < 6; ++t) d.E[t] = *reinterpret_cast<const T*>(eB + l_e[t]);
        wB += wstep; l_so += ss; l_o0 += sL; l_z += zsL; l_h += 1;
#pragma unroll
        for (int t = 0; t < 6; ++t) l_e[t] += es[t];
    };
    u32 st_so = so_base, st_o0 = o0_base;
    T zprev = Zero<T>::v();
    double zc0 = *reinterpret_cast<const double*>(zB + zo0), zc1 = *reinterpret_cast<const double*>(zB + zo1);
    double ihl0 = hB[0];
    T z0last = Zero<T>::v();
    auto fwd_step = [&](int i, const QFwd<T>& cur) {
        const bool lastb = (i == nL - 1);
        const double kL0 = 0.5 * ihl0, kL1 = 0.5 * cur.ihl1;
        const double rs0 = zc0 + zc1, rs1 = cur.n0 + cur.n1;
        const double cs0 = zc0 + cur.n0, cs1 = zc1 + cur.n1;
        T y = cur.S;
        cmac(y, cur.E[0], (Kc[0] * kL1) * rs1);
        cmac(y, cur.E[1], (Kc[1] * kL0) * rs0);
        cmac(y, cur.E[2], Kc[2] * cs1);
        cmac(y, cur.E[3], Kc[3] * cs0);
        cmac(y, cur.E[4], Kc[4] * cs1);
        cmac(y, cur.E[5], Kc[5] * cs0);
        const double cz = rs0 * ihl0;
        // coupling to the previous block (zprev = 0 at i = 0): row k: -d_k z_k with d_k = -kL0 cz;
        // row 0: -sum_k a_k z_k with a_k = ca cz, folded into the quad sum of its right-hand side terms
        cmac(y, zprev, kL0 * cz);
        if (lastb) y = Zero<T>::v();
        T part = cur.E[1] * (K0 * rs0);
        cmsc(part, zprev, ca * cz);
        const T y0 = cur.S0 + quad_add(part);
        const T y1 = quad_rot<1>(y), y2 = quad_rot<2>(y), y3 = quad_rot<3>(y);
        T z = cur.W[0] * y0;
        cmac(z, cur.W[1], y);
        cmac(z, cur.W[2], y1);
        cmac(z, cur.W[3], y2);
        cmac(z, cur.W[4], y3);
        // z_0 = W00 y0 + sum_k W[0][k] y_k  (W[0][k] = W[k][0]: the lane's cur.W[0]); off the chain
        T z0 = cur.W00 * y0 + quad_add(cur.W[0] * y);
        // park z_i in the unknowns themselves (overwritten by the backward pass; no other line of this
        // launch reads them)
        if (!lastb) *reinterpret_cast<T*>(eW + st_so) = z;
        if (k == 0) *reinterpret_cast<T*>(eW + st_o0) = z0;
        st_so += ss; st_o0 += sL;
        zprev = z;
        z0last = z0;
        zc0 = cur.n0; zc1 = cur.n1; ihl0 = cur.ihl1;
    };
    if (STAGES == 3) {
        QFwd<T> bA, bB, bC;
        load_fwd(0, bA);
        if (nL > 1) load_fwd(1, bB);
        int i = 0;
        for (; i + 3 <= nL - 2; i += 3) {
            load_fwd(i + 2, bC);
            fwd_step(i, bA);
            load_fwd(i + 3, bA);
            fwd_step(i + 1, bB);
            load_fwd(i + 4, bB);
            fwd_step(i + 2, bC);
        }
        // tail: blocks i .. nL-1 (at most 4 left; bA = block i, bB = block i+1 when it exists)
        if (i < nL) {
            if (i + 2 < nL) load_fwd(i + 2, bC);
            fwd_step(i, bA);
            if (i + 1 < nL) {
                if (i + 3 < nL) load_fwd(i + 3, bA);
                fwd_step(i + 1, bB);
                if (i + 2 < nL) {
                    fwd_step(i + 2, bC);
                    if (i + 3 < nL) fwd_step(i + 3, bA);
                }
            }
        }
    } else {
        QFwd<T> bA, bB;
        load_fwd(0, bA);
        int i = 0;
        for (; i + 1 <= nL - 1; i += 2) {
            load_fwd(i + 1, bB);
            fwd_step(i, bA);
            if (i + 2 < nL) load_fwd(i + 2, bA);
            fwd_step(i + 1, bB);
        }
        if (i < nL) fwd_step(i, bA);
    }

    // ----------------------------- backward --------------------------------
    // x_{nL-1} = z_{nL-1} (one unknown, already in place).  X0 = x_{i+1}[0] lives in every lane of the quad.
    if (nL < 2) return;
    T X0 = z0last;
    T xprev = Zero<T>::v();
    // cursors of block nL-2, descending
    const char* qW = reinterpret_cast<const char*>(a.fac) + (i64)(nL - 2) * wstep;
    u32 q_so = so_base + (u32)(nL - 2) * ss, q_o0 = o0_base + (u32)(nL - 2) * sL, q_z = (u32)(nL - 1) * zsL;
    const double* q_h = hB + (nL - 1);
    auto load_bwd = [&](int i, QBwd<T>& d) {
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(qW + wo[c]);
        d.zk = *reinterpret_cast<const T*>(eB + q_so);
        d.z0 = *reinterpret_cast<const T*>(eB + q_o0);
        d.p0 = *reinterpret_cast<const double*>(zB + (q_z + zo0));          // zeta pair at cell i+1
        d.p1 = *reinterpret_cast<const double*>(zB + (q_z + zo1));
        d.ihn = *q_h;
        qW -= wstep; q_so -= ss; q_o0 -= sL; q_z -= zsL; q_h -= 1;
    };
    u32 sq_so = so_base + (u32)(nL - 2) * ss, sq_o0 = o0_base + (u32)(nL - 2) * sL;
    auto bwd_step = [&](int i, const QBwd<T>& bc) {
        const double cz = (bc.p0 + bc.p1) * bc.ihn;
        const double ak = ca * cz;
        const double dk = (i + 1 == nL - 1) ? 0.0 : (-0.5 * bc.ihn) * cz;   // the last block has no d-coupling
        // v = A_{i+1}^T x_{i+1}: v_0 = 0, v_k = a_k x_0 + d_k x_k
        T v = X0 * ak;
        cmac(v, xprev, dk);
        const T v1 = quad_rot<1>(v), v2 = quad_rot<2>(v), v3 = quad_rot<3>(v);
        T x = bc.zk;
        cmsc(x, bc.W[1], v);
        cmsc(x, bc.W[2], v1);
        cmsc(x, bc.W[3], v2);
        cmsc(x, bc.W[4], v3);
        const T x0 = bc.z0 - quad_add(bc.W[0] * v);
        *reinterpret_cast<T*>(eW + sq_so) = x;
        if (k == 0) *reinterpret_cast<T*>(eW + sq_o0) = x0;
        sq_so -= ss; sq_o0 -= sL;
        X0 = x0;
        xprev = x;
    };
    if (STAGES == 3) {
        QBwd<T> bA, bB, bC;
        int i = nL - 2;
        load_bwd(i, bA);
        if (i >= 1) load_bwd(i - 1, bB);
        for (; i - 4 >= 0; i -= 3) {
            load_bwd(i - 2, bC);
            bwd_step(i, bA);
            load_bwd(i - 3, bA);
            bwd_step(i - 1, bB);
            load_bwd(i - 4, bB);
            bwd_step(i - 2, bC);
        }
        // tail: blocks i .. 0 (at most 4 left)
        if (i >= 0) {
            if (i - 2 >= 0) load_bwd(i - 2, bC);
            bwd_step(i, bA);
            if (i - 1 >= 0) {
                if (i - 3 >= 0) load_bwd(i - 3, bA);
                bwd_step(i - 1, bB);
                if (i - 2 >= 0) {
                    bwd_step(i - 2, bC);
                    if (i - 3 >= 0) bwd_step(i - 3, bA);
                }
            }
        }
    } else {
        QBwd<T> bA, bB;
        int i = nL - 2;
        load_bwd(i, bA);
        for (; i - 1 >= 0; i -= 2) {
            load_bwd(i - 1, bB);
            bwd_step(i, bA);
            if (i - 2 >= 0) load_bwd(i - 2, bA);
            bwd_step(i - 1, bB);
        }
        if (i >= 0) bwd_step(i, bA);
    }
}
