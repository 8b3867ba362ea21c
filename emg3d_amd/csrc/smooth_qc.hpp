// Quad-per-line chain kernel on the COMPACT factor: FOUR lanes per line, every exchange inside the quad done with DPP (no
// LDS, no barrier), 11 instead of 15 cached numbers per block (round 2's k_line_sweep_q on the full factor is in the git
// history).  Same recurrences, same elimination order (the reference's: emg3d/core.py:1447-1582 band LDL^T in block
// form, l_i before T_i inside a block):
//     forward : z_i = W_i (b_i - A_i z_{i-1})           backward: x_i = z_i - W_i A_{i+1}^T x_{i+1}
//
// Why: on bandwidth-saturated launches (256^3 level 0: 16 k lines, ~1 wave per SIMD, 4.9 TB/s of counted traffic, 13 % of
// the vector issue slots used) the factor is 45 % of the bytes -- 240 B per block in the forward and 224 B in the
// backward pass.  Of W_i = S_i^{-1} only the trailing block G_i = W_i[1..4][1..4] (10 numbers) and the first pivot's
// reciprocal r_i = 1 / S_i[0][0] are kept; column 0 of W_i is rebuilt in the kernel from what is in registers anyway:
//     S_i[T][0] = M_i[T][0] - D_i G_{i-1} a_i = -(a_i + D_i G_{i-1} a_i)        (M_i[T][0] = -a_i: the same zeta face and 1/h)
//     u_i = r_i S_i[T][0],   g_i = G_i u_i,   W_i[T][0] = -g_i,   W_i[0][0] = r_i + u_i . g_i
// (block elimination of the first unknown; a_i, D_i = row 0 / diagonal of the real coupling block A_i).  G_{i-1} is the
// previous block's row in the forward pass and the NEXT buffer of the register prefetch in the backward pass, a_i / D_i
// come from the zeta pair the lane holds already.  ~100 more instructions per block (4 c x r + 5 complex MACs and 18
// quad moves per pass, all off the dependent chain) for 176 + 176 instead of 240 + 224 bytes.
//
// z_0 is formed as r y_0 - u . z_T (one quad sum) instead of W_00 y_0 + W_0T . y_T (two).
//
// ZS (zeta separable): without magnetic permeabilities zeta IS the cell volume, zeta[i,j,k] = (hx_i hy_j) hz_k -- the
// product numpy forms in the reference's TensorMesh.cell_volumes (emg3d/meshes.py:140-147), which VolumeModel hands on
// unchanged (models.py:653-658).  The handle checks that bit for bit when it is created (k_zeta_is_volume); the kernel then
// forms the zeta pairs from the three width vectors (two multiplications in the reference's order) instead of reading
// them: 64 of the ~900 counted bytes per block.  Models with mu_r, and every coarse level (sums of fine zetas), read zeta.
//
// Source-free lines: the level-0 source of a survey is a dipole -- zero on all but a handful of lines.  The handle keeps one
// flag per line ("some source entry of this line is not +0", k_source_line_flags, recomputed whenever the source changes;
// LineArgs::sflag); a wave whose lines all have the flag clear runs a copy of the forward loop without the two source
// loads per lane and block (80 of the ~840 counted bytes per block).  The arithmetic is the same (y = 0 + ...): results are
// bit-identical.  Dense right-hand sides (Krylov vectors, every coarse level) have no flags and read their source.
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
// 16 lines per wave, all 64 lanes active (the lane-group kernels use 40 of 64), a dependent chain of two DPP stages instead of an LDS round trip.
//
// Why one-sided: round 1's plain two-sided elimination was 10^3-10^4 x less accurate on the
// ill-conditioned lines of the benchmark models (lines inside a resistive body: every interior node of a line
// carries a discrete gradient, a null vector of the curl-curl part that only eta regularises; condition
// ~ 1 / (omega mu sigma h^2) ~ 1e4..1e6).  Measured on the 128^3 model (tests/tools/conditioning.py, error of
// ONE line solve against 80-bit arithmetic): reference order 2e-12, two-sided 1e-8.  This kernel keeps the
// reference's elimination order, so a sweep agrees with the reference to ~1e-12 at every size.
#pragma once
#include <type_traits>
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

#ifndef EMG_Q_BLOCK
#define EMG_Q_BLOCK 256
#endif

template <class T>
struct QcFwd {          // what a lane loads for one forward block step
    T G[4];             // its row of G_i: [0] diagonal, [1..3] the other columns in rotation order
    T r;                // 1 / S_i[0][0]
    T S, S0;            // source of the row's own edge / of the edge along the line
    T E[6];
    double n0, n1;      // zeta pair of the row's side at cell i+1 (ZS: n0 = hL[i+1], the pair is formed from the widths)
    double ihl1;        // 1 / hL[i+1]
};
template <class T>
struct QcBwd {
    T G[4];
    T r;
    T zk, z0;
    double p0, p1, ihn; // zeta pair at cell i+1 (ZS: p0 = hL[i+1]), 1 / hL[i+1]
};

__device__ __forceinline__ double qc_rmul(double r, double s) { return r * s; }
__device__ __forceinline__ c128 qc_rmul(c128 r, c128 s) { return r * s; }

// BIG: field arrays of 4 GiB and more (512^3 complex: 6.4 GB) -- the per-lane byte offsets into e and s are 64 bits wide (a
// register pair and an add-with-carry per stream and step); everything else (factor planes, zeta, widths) is as before.
template <class T, int STAGES, int LPW, bool ZS = false, bool BIG = false>
__global__ __launch_bounds__(EMG_Q_BLOCK) void k_line_sweep_qc(LineArgs<T> a) {
    typedef unsigned int u32;
    typedef typename std::conditional<BIG, unsigned long long, u32>::type uof;     // byte offset into a field array
    const int lane = threadIdx.x & 63;
    const int k = lane & 3;                         // row k + 1
    const int g = lane >> 2;                        // line of the wave
    // lines per wave: the instantiation's, or fewer when the launch asks for it -- a launch of one-wave-per-SIMD waves lasts as
    // long as the SIMD with the most waves, so its lines are dealt evenly over the SIMDs instead of 16 at a time (MG::q_balanced_lpw)
    const int lpw = (a.qlpw > 0 && a.qlpw < LPW) ? a.qlpw : LPW;
    if (g >= lpw) return;
    EMG_SWEEP_WG(a)
    const i64 gidx = ((wg * blockDim.x + threadIdx.x) >> 6) * lpw + g;
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
    // ZS: the widths of the two cells of the row's zeta pair (element 0 / 1) across the line, and the product rule
    // zeta = (hx hy) hz in terms of (L, P, Q): z-lines (L = 2) (hP hQ) hL, x- and y-lines (hP hL) hQ
    const double* const wP = a.h[P];
    const double* const wQ = a.h[Q];
    const double hPa = ZS ? (tp ? wP[jPm + side] : wP[jPm]) : 0.0, hPb = ZS ? (tp ? wP[jPm + side] : wP[jP]) : 0.0;
    const double hQa = ZS ? (tp ? wQ[jQm] : wQ[jQm + side]) : 0.0, hQb = ZS ? (tp ? wQ[jQ] : wQ[jQm + side]) : 0.0;
    const bool zl2 = (L == 2);
    const double cPQa = hPa * hQa, cPQb = hPb * hQb;
    // (the empty asm keeps the rounded product apart from the additions it feeds: fused into an FMA it would differ from the
    // stored zeta in the last bit)
    auto rounded = [](double x) -> double { asm volatile("" : "+v"(x)); return x; };
    auto zeta_a = [&](double hl) -> double { return rounded(zl2 ? cPQa * hl : (hPa * hl) * hQa); };
    auto zeta_b = [&](double hl) -> double { return rounded(zl2 ? cPQb * hl : (hPb * hl) * hQb); };
    const i64 o0 = FL_(0, jP, jQ);                  // the edge along the line, block 0
#undef FL_
#undef FP_
#undef FQ_
#undef SPC_
#undef SPN_

    // Addressing as in k_line_sweep_q: uniform base pointers + 32-bit per-lane BYTE offsets.
    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    char* const eW = reinterpret_cast<char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const double* const hB = a.ih[L];
    const double* const wL = a.h[L];
    const i64 wstep = 11 * nLt * (i64)sizeof(T);
    u32 wo[4];                                       // entry offsets of the lane's row of G inside one block record
    {
        const int cols[4] = {rr, 1 + ((k + 1) & 3), 1 + ((k + 2) & 3), 1 + ((k + 3) & 3)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r1 = rr > cols[c] ? rr : cols[c], c1 = rr > cols[c] ? cols[c] : rr;
            wo[c] = (u32)(((i64)((r1 - 1) * r1 / 2 + (c1 - 1)) * nLt + slot) * (i64)sizeof(T));
        }
    }
    const u32 wr = (u32)(((i64)10 * nLt + slot) * (i64)sizeof(T));
    const uof ss = (uof)(os[0] * (i64)sizeof(T));               // stride of the row's own edge
    const uof sL = (uof)(fl.st[L][L] * (i64)sizeof(T));         // stride of the edge along the line
    const uof so_base = (uof)(ob[0] * (i64)sizeof(T));
    const uof o0_base = (uof)(o0 * (i64)sizeof(T));
    uof es[6], eb_[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { eb_[t] = (uof)(ob[1 + t] * (i64)sizeof(T)); es[t] = (uof)(os[1 + t] * (i64)sizeof(T)); }
    const u32 zo0 = (u32)(fb * 8), zo1 = (u32)((fb + sv) * 8), zsL = (u32)(csL * 8);

    // column 0 of W_i from (G_{i-1} row, G_i row, r_i, a_k, d_k of block i): u_k (out) and g_k (returned)
    auto col0 = [&](const T Gp[4], const T G[4], const T r, double ak, double dk, T& u) -> T {
        const double a1 = quad_rot<1>(ak), a2 = quad_rot<2>(ak), a3 = quad_rot<3>(ak);
        T Ga = Gp[0] * ak;
        cmac(Ga, Gp[1], a1); cmac(Ga, Gp[2], a2); cmac(Ga, Gp[3], a3);
        T sk = Ga * dk;
        add_real(sk, ak);                            // a_k + d_k (G' a)_k  =  -S_i[k][0]
        u = -qc_rmul(r, sk);
        const T u1 = quad_rot<1>(u), u2 = quad_rot<2>(u), u3 = quad_rot<3>(u);
        T gk = G[0] * u;
        cmac(gk, G[1], u1); cmac(gk, G[2], u2); cmac(gk, G[3], u3);
        return gk;
    };

    // ----------------------------- forward ---------------------------------
    const char* wB = reinterpret_cast<const char*>(a.fac);
    uof l_so = so_base, l_o0 = o0_base, l_e[6];
    u32 l_z = zsL;
#pragma unroll
    for (int t = 0; t < 6; ++t) l_e[t] = eb_[t];
    const double* l_h = hB + 1;
    auto load_fwd = [&](int i, QcFwd<T>& d, auto nosrc_) {
        const bool lastb = (i == nL - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) d.G[c] = *(reinterpret_cast<const T*>(wB + wo[c]));
        d.r = *(reinterpret_cast<const T*>(wB + wr));
        if (ZS) {
            d.n0 = wL[lastb ? i : i + 1];
        } else {
            d.n0 = *reinterpret_cast<const double*>(zB + ((lastb ? l_z - zsL : l_z) + zo0));
            d.n1 = *reinterpret_cast<const double*>(zB + ((lastb ? l_z - zsL : l_z) + zo1));
        }
        d.ihl1 = lastb ? l_h[-1] : l_h[0];
        if constexpr (decltype(nosrc_)::value) {
            d.S = Zero<T>::v(); d.S0 = Zero<T>::v();
        } else {
            d.S = *(reinterpret_cast<const T*>(sB + l_so));
            d.S0 = *(reinterpret_cast<const T*>(sB + l_o0));
        }
        d.E[0] = *(reinterpret_cast<const T*>(eB + (lastb ? l_e[0] - es[0] : l_e[0])));
#pragma unroll
        for (int t = 1; t < 6; ++t) d.E[t] = *(reinterpret_cast<const T*>(eB + l_e[t]));
        wB += wstep; l_so += ss; l_o0 += sL; l_z += zsL; l_h += 1;
#pragma unroll
        for (int t = 0; t < 6; ++t) l_e[t] += es[t];
    };
    uof st_so = so_base, st_o0 = o0_base;
    T zprev = Zero<T>::v();
    T Gp[4] = {Zero<T>::v(), Zero<T>::v(), Zero<T>::v(), Zero<T>::v()};      // row of G_{i-1} (G_{-1} = 0)
    double zc0 = ZS ? zeta_a(wL[0]) : *reinterpret_cast<const double*>(zB + zo0);
    double zc1 = ZS ? zeta_b(wL[0]) : *reinterpret_cast<const double*>(zB + zo1);
    double ihl0 = hB[0];
    T z0last = Zero<T>::v();
    auto fwd_step = [&](int i, const QcFwd<T>& cur) {
        const bool lastb = (i == nL - 1);
        const double kL0 = 0.5 * ihl0, kL1 = 0.5 * cur.ihl1;
        const double cn0 = ZS ? zeta_a(cur.n0) : cur.n0, cn1 = ZS ? zeta_b(cur.n0) : cur.n1;
        const double rs0 = zc0 + zc1, rs1 = cn0 + cn1;
        const double cs0 = zc0 + cn0, cs1 = zc1 + cn1;
        T y = cur.S;
        cmac(y, cur.E[0], (Kc[0] * kL1) * rs1);
        cmac(y, cur.E[1], (Kc[1] * kL0) * rs0);
        cmac(y, cur.E[2], Kc[2] * cs1);
        cmac(y, cur.E[3], Kc[3] * cs0);
        cmac(y, cur.E[4], Kc[4] * cs1);
        cmac(y, cur.E[5], Kc[5] * cs0);
        const double cz = rs0 * ihl0;
        const double ak = ca * cz, dk = -kL0 * cz;       // A_i: row 0 = a_k, diagonal = d_k
        // off the chain: column 0 of W_i
        T u;
        const T gk = col0(Gp, cur.G, cur.r, ak, dk, u);
        // coupling to the previous block (zprev = 0 at i = 0): row k: -d_k z_k; row 0: -sum_k a_k z_k (quad sum)
        cmsc(y, zprev, dk);
        if (lastb) y = Zero<T>::v();
        T part = cur.E[1] * (K0 * rs0);
        cmsc(part, zprev, ak);
        const T y0 = cur.S0 + quad_add(part);
        const T y1 = quad_rot<1>(y), y2 = quad_rot<2>(y), y3 = quad_rot<3>(y);
        T z = cur.G[0] * y;
        cmac(z, cur.G[1], y1);
        cmac(z, cur.G[2], y2);
        cmac(z, cur.G[3], y3);
        cmsc(z, gk, y0);                                 // W_i[k][0] = -g_k
        // z_0 = r y_0 - u . z_T; off the chain
        const T z0 = qc_rmul(cur.r, y0) - quad_add(u * z);
        if (!lastb) *reinterpret_cast<T*>(eW + st_so) = z;
        if (k == 0) *reinterpret_cast<T*>(eW + st_o0) = z0;
        st_so += ss; st_o0 += sL;
        zprev = z;
        z0last = z0;
#pragma unroll
        for (int c = 0; c < 4; ++c) Gp[c] = cur.G[c];
        zc0 = cn0; zc1 = cn1; ihl0 = cur.ihl1;
    };
    // all lines of the wave source-free (wave-uniform): the loop copy without source loads
    const bool nosrc = a.sflag != nullptr &&
                       __builtin_amdgcn_ballot_w64(a.sflag[(i64)bsys_ * a.nLinesTot + slot] == 0) == __builtin_amdgcn_ballot_w64(true);
    auto forward = [&](auto ns) {
    if (STAGES == 3) {
        QcFwd<T> bA, bB, bC;
        load_fwd(0, bA, ns);
        if (nL > 1) load_fwd(1, bB, ns);
        int i = 0;
        for (; i + 3 <= nL - 2; i += 3) {
            load_fwd(i + 2, bC, ns);
            fwd_step(i, bA);
            load_fwd(i + 3, bA, ns);
            fwd_step(i + 1, bB);
            load_fwd(i + 4, bB, ns);
            fwd_step(i + 2, bC);
        }
        if (i < nL) {
            if (i + 2 < nL) load_fwd(i + 2, bC, ns);
            fwd_step(i, bA);
            if (i + 1 < nL) {
                if (i + 3 < nL) load_fwd(i + 3, bA, ns);
                fwd_step(i + 1, bB);
                if (i + 2 < nL) {
                    fwd_step(i + 2, bC);
                    if (i + 3 < nL) fwd_step(i + 3, bA);
                }
            }
        }
    } else {
        QcFwd<T> bA, bB;
        load_fwd(0, bA, ns);
        int i = 0;
        for (; i + 1 <= nL - 1; i += 2) {
            load_fwd(i + 1, bB, ns);
            fwd_step(i, bA);
            if (i + 2 < nL) load_fwd(i + 2, bA, ns);
            fwd_step(i + 1, bB);
        }
        if (i < nL) fwd_step(i, bA);
    }
    };
    if (nosrc) forward(std::true_type{}); else forward(std::false_type{});

    // ----------------------------- backward --------------------------------
    // x_{nL-1} = z_{nL-1} (one unknown, already in place).  X0 = x_{i+1}[0] lives in every lane of the quad.
    // Step i uses its own buffer (G_i, r_i, z_i, the zeta pair of cell i+1: coupling A_{i+1}) and the NEXT buffer of the
    // prefetch (block i-1: G_{i-1} and the zeta pair of cell i, i.e. a_i, d_i) for column 0 of W_i.
    if (nL < 2) return;
    T X0 = z0last;
    T xprev = Zero<T>::v();
    const char* qW = reinterpret_cast<const char*>(a.fac) + (i64)(nL - 2) * wstep;
    uof q_so = so_base + (uof)(nL - 2) * ss, q_o0 = o0_base + (uof)(nL - 2) * sL;
    u32 q_z = (u32)(nL - 1) * zsL;
    const double* q_h = hB + (nL - 1);
    auto load_bwd = [&](int i, QcBwd<T>& d) {        // i = -1: only the zeta pair / width of cell 0 (G_{-1} = 0)
        if (i >= 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) d.G[c] = *(reinterpret_cast<const T*>(qW + wo[c]));
            d.r = *(reinterpret_cast<const T*>(qW + wr));
            d.zk = *(reinterpret_cast<const T*>(eB + q_so));
            d.z0 = *(reinterpret_cast<const T*>(eB + q_o0));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) d.G[c] = Zero<T>::v();
            d.r = Zero<T>::v(); d.zk = Zero<T>::v(); d.z0 = Zero<T>::v();
        }
        if (ZS) {
            d.p0 = wL[i + 1];                                                   // cell i+1 (i = -1: cell 0)
        } else {
            d.p0 = *reinterpret_cast<const double*>(zB + (q_z + zo0));          // zeta pair at cell i+1
            d.p1 = *reinterpret_cast<const double*>(zB + (q_z + zo1));
        }
        d.ihn = *q_h;
        qW -= wstep; q_so -= ss; q_o0 -= sL; q_z -= zsL; q_h -= 1;
    };
    uof sq_so = so_base + (uof)(nL - 2) * ss, sq_o0 = o0_base + (uof)(nL - 2) * sL;
    auto bwd_step = [&](int i, const QcBwd<T>& bc, const QcBwd<T>& nx) {
        const double cz = (ZS ? zeta_a(bc.p0) + zeta_b(bc.p0) : bc.p0 + bc.p1) * bc.ihn;
        const double ak = ca * cz;
        const double dk = (i + 1 == nL - 1) ? 0.0 : (-0.5 * bc.ihn) * cz;   // the last block has no d-coupling
        // off the chain: column 0 of W_i from block i's own coupling coefficients (cell i: the next buffer's zeta pair)
        const double czi = (ZS ? zeta_a(nx.p0) + zeta_b(nx.p0) : nx.p0 + nx.p1) * nx.ihn;
        T u;
        const T gk = col0(nx.G, bc.G, bc.r, ca * czi, (-0.5 * nx.ihn) * czi, u);
        // v = A_{i+1}^T x_{i+1}: v_0 = 0, v_k = a_k x_0 + d_k x_k
        T v = X0 * ak;
        cmac(v, xprev, dk);
        const T v1 = quad_rot<1>(v), v2 = quad_rot<2>(v), v3 = quad_rot<3>(v);
        T x = bc.zk;
        cmsc(x, bc.G[0], v);
        cmsc(x, bc.G[1], v1);
        cmsc(x, bc.G[2], v2);
        cmsc(x, bc.G[3], v3);
        const T x0 = bc.z0 + quad_add(gk * v);           // - W_i[0][T] . v,  W_i[0][k] = -g_k
        *reinterpret_cast<T*>(eW + sq_so) = x;
        if (k == 0) *reinterpret_cast<T*>(eW + sq_o0) = x0;
        sq_so -= ss; sq_o0 -= sL;
        X0 = x0;
        xprev = x;
    };
    if (STAGES == 3) {
        QcBwd<T> bA, bB, bC;
        int i = nL - 2;
        load_bwd(i, bA);
        load_bwd(i - 1, bB);
        for (; i - 3 >= 0; i -= 3) {                 // (blocks i-3 .. i-4 >= -1 exist as buffers)
            load_bwd(i - 2, bC);
            bwd_step(i, bA, bB);
            load_bwd(i - 3, bA);
            bwd_step(i - 1, bB, bC);
            load_bwd(i - 4, bB);
            bwd_step(i - 2, bC, bA);
        }
        // tail: blocks i .. 0 (at most 3 left); bA = block i, bB = block i-1 (or the cell-0 stub)
        if (i >= 0) {
            if (i - 1 >= 0) load_bwd(i - 2, bC);
            bwd_step(i, bA, bB);
            if (i - 1 >= 0) {
                if (i - 2 >= 0) load_bwd(i - 3, bA);
                bwd_step(i - 1, bB, bC);
                if (i - 2 >= 0) bwd_step(i - 2, bC, bA);
            }
        }
    } else {
        QcBwd<T> bA, bB;
        int i = nL - 2;
        load_bwd(i, bA);
        for (; i - 1 >= 0; i -= 2) {
            load_bwd(i - 1, bB);
            bwd_step(i, bA, bB);
            load_bwd(i - 2, bA);
            bwd_step(i - 1, bB, bA);
        }
        if (i >= 0) { load_bwd(i - 1, bB); bwd_step(i, bA, bB); }
    }
}
