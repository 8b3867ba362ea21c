// Line sweep staged in LDS, for the levels whose colour launches are bound by the length of a chain step rather than by
// bytes (lines of 16 .. 128 blocks, a few thousand lines per colour; profiles/HISTORY.md A.13).
//
// The row-parallel chain kernels (k_line_sweep_rp / _thm) spend ~270 instructions per block step, ~200 of them on work that
// does not depend on the previous block: loading and assembling the right-hand side from the source and the 12 neighbour
// values.  A wave runs them one after the other, 4-5 cycles each, for every block of its lines in turn.  Here a workgroup of
// 256 threads takes LPW lines and
//   phase A (all threads, nothing depends on anything): the factor rows of every block of its lines -> LDS; the right-hand
//           side b_r and the coupling coefficient cz_r of every (block, row, line) -> LDS;
//   phase B (wave 0): the chain itself -- z_i = W_i (b_i - A_i z_{i-1}) forwards, x_i = z_i - W_i A_{i+1}^T x_{i+1}
//           backwards -- with everything it reads in LDS: ~60 instructions per step; z is parked in LDS, the factor is read
//           from HBM once, x is stored once.
// The arithmetic is k_line_sweep_rp's, statement by statement (one-sided block factorisation [block][15][line] of
// k_line_factor, mid = nL - 1): same results to rounding (1e-11: the multiply-adds are contracted differently).
// MEASURED (MI355X, 64 x 128 x 64, profiles/HISTORY.md A.14): a chain step is ~110 instructions here against 273 in the
// two-sided kernel, but takes ~0.3 us against 0.55 (one exposed LDS round trip per step, in-order issue), and the
// one-sided chain has twice the steps: ~46 us per round of 64-block lines (two rounds at 7 lines per workgroup) against 35
// (two-sided chain) / 40 (scan kernel).  Lab build only, off by default (EMG3D_LDS=1); a two-sided version is the open candidate.
// Dynamic LDS per line and block: 5 b + 15 W entries + 4 cz = 20 T + 4 doubles (352 B complex, 192 B real); the host picks
// LPW = lines per workgroup so that nL * LPW of them (+ 8 B per block for 1 / hL) fit 156 KB.  Colour ordering only (mode 0).
#pragma once
#include "smooth.hpp"

#define EMG_LDS_BLOCK 256
#define EMG_LDS_BYTES (156 * 1024)

template <class T>
HD size_t lds_bytes_per_line_block() { return 20 * sizeof(T) + 4 * sizeof(double); }

// Per-row configuration of a lane (the table of k_line_sweep_rp): row r (0: the line's own edge, 1 / 2: the P edges at the
// lower / upper side, 3 / 4: the Q edges) of the line (jP, jQ).
template <class T>
struct LdsRow {
    typedef unsigned int u32;
    u32 so, ss;             // the row's own unknown: byte offset at block 0, per-block stride
    u32 eo[6], es[6];       // the six neighbour values
    u32 zo0, zo1, zsu;      // zeta face
    double K[6];
    double ca, tmask;
    bool t0;
};

template <class T>
__device__ __forceinline__ void lds_row_setup(const LineArgs<T>& a, int rr, i64 jP, i64 jQ, LdsRow<T>& c) {
    typedef unsigned int u32;
    const int L = a.L, P = a.P, Q = a.Q;
    const i64 csP = a.cl.st[P], csQ = a.cl.st[Q];
    const double ihP[2] = {a.ih[P][jP - 1], a.ih[P][jP]};
    const double ihQ[2] = {a.ih[Q][jQ - 1], a.ih[Q][jQ]};
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    const FieldLayout& fl = a.fl;
    const i64 jPm = jP - 1, jPp = jP + 1, jQm = jQ - 1, jQp = jQ + 1;
#define FL_(vL, vP, vQ) (fl.off[L] + (vL) * fl.st[L][L] + (vP) * fl.st[L][P] + (vQ) * fl.st[L][Q])
#define FP_(vL, vP, vQ) (fl.off[P] + (vL) * fl.st[P][L] + (vP) * fl.st[P][P] + (vQ) * fl.st[P][Q])
#define FQ_(vL, vP, vQ) (fl.off[Q] + (vL) * fl.st[Q][L] + (vP) * fl.st[Q][P] + (vQ) * fl.st[Q][Q])
    const i64 cP0 = (jP - 1) * csP, cP1 = jP * csP, cq = (jQ - 1) * csQ;
    const int type = (rr == 0) ? 0 : (rr <= 2 ? 1 : 2);
    const int side = (rr == 0) ? 0 : ((rr - 1) & 1);
    const double sg = side ? -1.0 : 1.0;
    i64 ob[7], os[7], fb, sv, suT0;
    c.ca = 0.0;
    c.tmask = (type == 0) ? 0.0 : 1.0;
    c.t0 = (type == 0);
    if (type == 0) {
        ob[0] = FL_(0, jP, jQ);
        ob[1] = FL_(0, jPp, jQ); ob[2] = FL_(0, jPm, jQ); ob[3] = FL_(0, jP, jQp); ob[4] = FL_(0, jP, jQm);
        ob[5] = ob[1]; ob[6] = ob[1];
#pragma unroll
        for (int t = 0; t < 7; ++t) os[t] = fl.st[L][L];
        fb = cP0 + cq; sv = csQ; suT0 = cP1 - cP0;
        c.K[0] = kP[1] * ihP[1]; c.K[1] = kP[0] * ihP[0]; c.K[2] = kQ[1] * ihQ[1]; c.K[3] = kQ[0] * ihQ[0];
        c.K[4] = 0.0; c.K[5] = 0.0;
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
        c.K[0] = sg * ihA; c.K[1] = -sg * ihA;
        c.K[2] = sg * kQ[1] * ihA; c.K[3] = -sg * kQ[0] * ihA;
        c.K[4] = kQ[1] * ihQ[1]; c.K[5] = kQ[0] * ihQ[0];
        c.ca = sg * 0.5 * ihA;
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
        c.K[0] = sg * ihA; c.K[1] = -sg * ihA;
        c.K[2] = sg * kP[1] * ihA; c.K[3] = -sg * kP[0] * ihA;
        c.K[4] = kP[1] * ihP[1]; c.K[5] = kP[0] * ihP[0];
        c.ca = sg * 0.5 * ihA;
    }
#undef FL_
#undef FP_
#undef FQ_
    const u32 TS = (u32)sizeof(T);
    c.so = (u32)(ob[0] * (i64)TS); c.ss = (u32)(os[0] * (i64)TS);
#pragma unroll
    for (int t = 0; t < 6; ++t) { c.eo[t] = (u32)(ob[1 + t] * (i64)TS); c.es[t] = (u32)(os[1 + t] * (i64)TS); }
    c.zo0 = (u32)(fb * 8); c.zo1 = (u32)((fb + sv) * 8); c.zsu = (u32)(suT0 * 8);
}

template <class T>
__global__ __launch_bounds__(EMG_LDS_BLOCK) void k_line_sweep_lds(LineArgs<T> a) {
    typedef unsigned int u32;
    const int LPW = a.lds;                      // lines of this workgroup
    EMG_SWEEP_WG(a)
    const i64 nlines = a.cntA * a.cntB;         // colour mode only
    const int nL = (int)a.nC[a.L];
    const i64 nLt = a.nLinesTot;
    const u32 csL8 = (u32)(a.cl.st[a.L] * 8);
    const u32 TS = (u32)sizeof(T);
    const char* const eB = reinterpret_cast<const char*>(a.e + boff_);
    char* const eW = reinterpret_cast<char*>(a.e + boff_);
    const char* const sB = reinterpret_cast<const char*>(a.s + boff_);
    const char* const wB = reinterpret_cast<const char*>(a.fac);
    const char* const zB = reinterpret_cast<const char*>(a.zeta);
    const double* const hB = a.ih[a.L];

    extern __shared__ __attribute__((aligned(16))) char lds_dyn[];
    const int RG = 5 * LPW;
    T* const yb = reinterpret_cast<T*>(lds_dyn);                    // [block][row 0..4][line]
    T* const wl = yb + (size_t)nL * RG;                             // [block][entry 0..14][line]
    double* const czb = reinterpret_cast<double*>(wl + (size_t)nL * 15 * LPW);     // [block][row 1..4][line]
    double* const ihs = czb + (size_t)nL * 4 * LPW;                 // 1 / hL of every block
    __shared__ T xch[2][64];

    // the line of slot g of this workgroup (clamped to the colour's last line: duplicates compute, never store)
    auto line_of = [&](int g, i64& jP, i64& jQ, bool& valid) {
        i64 gidx = wg * LPW + g;
        valid = gidx < nlines;
        if (!valid) gidx = nlines - 1;
        const i64 b = gidx / a.cntA, qq = gidx - b * a.cntA;
        jP = 1 + a.cP + 2 * qq;
        jQ = 1 + a.cQ + 2 * b;
    };
    if (wg * LPW >= nlines) return;             // whole workgroup past the end (XCD map padding)

    // ---------------------------------------------------------------- phase A
    // (every loop below issues the loads of several iterations before it uses the first: nothing here depends on
    // anything, the only cost is one memory round trip per batch)
    {
        // factor: entry p of block i of line g -- thread -> line g fixed, (block, entry) pairs ip0, ip0 + NP, ...;
        // consecutive threads take consecutive lines (contiguous slots)
        const int NP = EMG_LDS_BLOCK / LPW;
        if ((int)threadIdx.x < NP * LPW) {
            const int g = threadIdx.x % LPW, ip0 = threadIdx.x / LPW;
            i64 jP, jQ; bool valid;
            line_of(g, jP, jQ, valid);
            const char* const src = wB + line_slot(a, jP, jQ) * (i64)TS;
            const i64 rowb = nLt * (i64)TS;
            const int nIP = nL * 15;
            auto ldw = [&](int q, int ip) -> T { return *reinterpret_cast<const T*>(src + (i64)(q < nIP ? q : ip) * rowb); };
            for (int ip = ip0; ip < nIP; ip += 6 * NP) {
                const int q1 = ip + NP, q2 = ip + 2 * NP, q3 = ip + 3 * NP, q4 = ip + 4 * NP, q5 = ip + 5 * NP;
                const T t0 = ldw(ip, ip), t1 = ldw(q1, ip), t2 = ldw(q2, ip), t3 = ldw(q3, ip), t4 = ldw(q4, ip), t5 = ldw(q5, ip);
                wl[(size_t)ip * LPW + g] = t0;
                if (q1 < nIP) wl[(size_t)q1 * LPW + g] = t1;
                if (q2 < nIP) wl[(size_t)q2 * LPW + g] = t2;
                if (q3 < nIP) wl[(size_t)q3 * LPW + g] = t3;
                if (q4 < nIP) wl[(size_t)q4 * LPW + g] = t4;
                if (q5 < nIP) wl[(size_t)q5 * LPW + g] = t5;
            }
        }
        for (int k = threadIdx.x; k < nL; k += EMG_LDS_BLOCK) ihs[k] = hB[k];
        // right-hand sides: thread -> (row, line) fixed, blocks grp, grp + NG, ...
        const int NG = EMG_LDS_BLOCK / RG;
        const int grp = threadIdx.x / RG, rg = threadIdx.x - grp * RG;
        if (grp < NG) {
            const int r = rg / LPW, g = rg - r * LPW;
            i64 jP, jQ; bool valid;
            line_of(g, jP, jQ, valid);
            LdsRow<T> c;
            lds_row_setup(a, r, jP, jQ, c);
            struct Ops { T S, E[6]; double zf0, zf1, zf2, zf3, ihl0, ihl1; };
            auto load_ops = [&](int i, Ops& o) {
                const bool lastb = (i == nL - 1);
                const u32 su = c.t0 ? c.zsu : (lastb ? 0u : csL8);
                const u32 zb = (u32)i * csL8;
                o.zf0 = *reinterpret_cast<const double*>(zB + (zb + c.zo0));
                o.zf1 = *reinterpret_cast<const double*>(zB + (zb + c.zo1));
                o.zf2 = *reinterpret_cast<const double*>(zB + (zb + c.zo0 + su));
                o.zf3 = *reinterpret_cast<const double*>(zB + (zb + c.zo1 + su));
                o.ihl0 = hB[i]; o.ihl1 = hB[lastb ? i : i + 1];
                const bool clamp = (!c.t0) && lastb;
                const u32 ii = (u32)(clamp ? i - 1 : i);
                o.S = *reinterpret_cast<const T*>(sB + (c.so + ii * c.ss));
#pragma unroll
                for (int t = 0; t < 6; ++t) o.E[t] = *reinterpret_cast<const T*>(eB + (c.eo[t] + ii * c.es[t]));
            };
            auto rhs_store = [&](int i, const Ops& o) {
                const double kL0 = 0.5 * o.ihl0, kL1 = 0.5 * o.ihl1;
                const double rs0 = o.zf0 + o.zf1, rs1 = o.zf2 + o.zf3;
                const double cs0 = o.zf0 + o.zf2, cs1 = o.zf1 + o.zf3;
                const double g0 = (c.t0 ? c.K[0] : c.K[0] * kL1) * rs1;
                const double g1 = (c.t0 ? c.K[1] : c.K[1] * kL0) * rs0;
                T y = o.S;
                y += g0 * o.E[0];
                y += g1 * o.E[1];
                y += (c.K[2] * cs1) * o.E[2];
                y += (c.K[3] * cs0) * o.E[3];
                y += (c.K[4] * cs1) * o.E[4];
                y += (c.K[5] * cs0) * o.E[5];
                yb[(size_t)i * RG + rg] = y;
                if (r > 0) czb[((size_t)i * 4 + (r - 1)) * LPW + g] = rs0 * o.ihl0;
            };
            for (int i = grp; i < nL; i += 2 * NG) {
                Ops o0, o1;
                const int i1 = i + NG;
                load_ops(i, o0);
                load_ops(i1 < nL ? i1 : i, o1);
                rhs_store(i, o0);
                if (i1 < nL) rhs_store(i1, o1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;

    // ---------------------------------------------------------------- phase B: the chain (wave 0)
    const int lane = threadIdx.x;
    const int r = lane / LPW;                   // >= 5: mirror lanes (no stores)
    const int g = lane - r * LPW;
    const bool rowact = r < 5;
    const int rr = rowact ? r : 0;
    i64 jP, jQ; bool valid;
    line_of(g, jP, jQ, valid);
    LdsRow<T> c;
    lds_row_setup(a, rr, jP, jQ, c);
    const bool store = rowact && valid;
    T* const xu = xch[0];
    T* const xy = xch[1];
    const int rgB = rr * LPW + g;
    int wi[5];
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) wi[cc] = wpk(rr, cc) * LPW + g;
    const int czi = (rr > 0 ? rr - 1 : 0) * LPW + g;
    const double czm = rr > 0 ? 1.0 : 0.0;      // row 0 has no coupling coefficient of its own
    struct Fw { T y, W[5]; double cz, ih; };
    struct Bw { T zi, W[5]; double cz, ih; };
    auto ld_f = [&](int i, Fw& d) {
        const T* const wrow = wl + (size_t)i * 15 * LPW;
        d.y = yb[(size_t)i * RG + rgB];
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) d.W[cc] = wrow[wi[cc]];
        d.cz = czb[(size_t)i * 4 * LPW + czi];
        d.ih = ihs[i];
    };
    auto ld_b = [&](int i, Bw& d) {             // block i >= 0; coefficients of block i + 1
        const T* const wrow = wl + (size_t)i * 15 * LPW;
        d.zi = yb[(size_t)i * RG + rgB];
#pragma unroll
        for (int cc = 1; cc < 5; ++cc) d.W[cc] = wrow[wi[cc]];
        d.cz = czb[(size_t)(i + 1) * 4 * LPW + czi];
        d.ih = ihs[i + 1];
    };
    T zprev = Zero<T>::v();
    auto fstep = [&](int i, const Fw& cur) {
        const bool full = c.t0 || i < nL - 1;
        const double kL0 = 0.5 * cur.ih;
        const double cz = czm * cur.cz;
        T y = cur.y;
        y += ((c.tmask * kL0) * cz) * zprev;
        if (!full) y = Zero<T>::v();
        xy[lane] = y;
        xu[lane] = (c.ca * cz) * zprev;
        const T y0 = xy[g], y1 = xy[LPW + g], y2 = xy[2 * LPW + g], y3 = xy[3 * LPW + g], y4 = xy[4 * LPW + g];
        const T u1 = xu[LPW + g], u2 = xu[2 * LPW + g], u3 = xu[3 * LPW + g], u4 = xu[4 * LPW + g];
        const T su = (u1 + u2) + (u3 + u4);
        const T z = ((cur.W[0] * (y0 - su) + cur.W[1] * y1) + (cur.W[2] * y2 + cur.W[3] * y3)) + cur.W[4] * y4;
        if (rowact) yb[(size_t)i * RG + rgB] = z;           // parked in LDS
        zprev = z;
    };
    {
        // two register sets, unrolled by two: the next block's operands are read while this one computes
        Fw A, B;
        ld_f(0, A);
        int i = 0;
        for (; i + 1 < nL; i += 2) {
            ld_f(i + 1, B);
            fstep(i, A);
            ld_f(i + 2 < nL ? i + 2 : i + 1, A);
            fstep(i + 1, B);
        }
        if (i < nL) fstep(i, A);
    }
    // x_{nL-1} = z_{nL-1} (only row 0 exists on the last block)
    if (store && c.t0) *reinterpret_cast<T*>(eW + (c.so + (u32)(nL - 1) * c.ss)) = zprev;
    auto bstep = [&](int i, const Bw& cur) {
        const double ihLn = cur.ih;
        const double cz = czm * cur.cz;
        const double dm = (i + 1 == nL - 1) ? 0.0 : c.tmask;
        const double ac = c.ca * cz;
        const double dc = ((-0.5 * dm) * ihLn) * cz;
        xy[lane] = c.t0 ? zprev : dc * zprev;
        T aa = Zero<T>::v();
        add_real(aa, ac);
        xu[lane] = aa;
        const T x0 = xy[g];
        const T v1 = real_of(xu[LPW + g]) * x0 + xy[LPW + g];
        const T v2 = real_of(xu[2 * LPW + g]) * x0 + xy[2 * LPW + g];
        const T v3 = real_of(xu[3 * LPW + g]) * x0 + xy[3 * LPW + g];
        const T v4 = real_of(xu[4 * LPW + g]) * x0 + xy[4 * LPW + g];
        const T w = (cur.W[1] * v1 + cur.W[2] * v2) + (cur.W[3] * v3 + cur.W[4] * v4);
        const T x = cur.zi - w;
        if (store) *reinterpret_cast<T*>(eW + (c.so + (u32)i * c.ss)) = x;
        zprev = x;
    };
    if (nL >= 2) {
        Bw A, B;
        ld_b(nL - 2, A);
        int i = nL - 2;
        for (; i >= 1; i -= 2) {
            ld_b(i - 1, B);
            bstep(i, A);
            ld_b(i >= 2 ? i - 2 : 0, A);
            bstep(i - 1, B);
        }
        if (i == 0) bstep(0, A);
    }
}
