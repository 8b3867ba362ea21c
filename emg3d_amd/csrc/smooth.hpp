// Gauss-Seidel smoothers (reference emg3d/core.py:181-1316).
//
// Line relaxation.  For a line along axis L at transverse node (jP, jQ) the
// reference assembles a complex-symmetric 11-band system of 5*nL-4 unknowns
// (block-tridiagonal, 5x5 blocks `middle`, sparse real coupling `left`,
// core.py:608-691) and solves it with a non-pivoting band LDL^T
// (core.py:1447-1582).  The matrix depends only on the model (eta, zeta, h),
// NOT on the field, so it is the same in every sweep of every cycle.  Here the
// band LDL^T is computed ONCE per (level, direction) in its block form
//     S_0 = M_0,   S_i = M_i - A_i S_{i-1}^{-1} A_i^T,   S_i = L_i D_i L_i^T
// (k_line_factor; 15 numbers per block: the symmetric explicit inverse
// W_i = S_i^{-1}, obtained from a non-pivoting LDL^T of S_i; the sub-diagonal
// coupling A_i is recomputed from zeta) and kept in HBM (288 GB make that affordable: 240 B per cell and
// direction).  A sweep (k_line_sweep) is then only
//     forward : y_i = b_i - A_i z_{i-1},  z_i = S_i^{-1} y_i
//     backward: x_i = z_i - S_i^{-1} A_{i+1}^T x_{i+1}
// with b_i the reference's right-hand side (core.py:697-736).  This is the
// same factorisation the reference computes (its scalar band LDL^T restricted
// to the block structure), so results agree to rounding.
//
// Parallelisation: one thread per line; lanes run over the transverse node
// index jP, the fastest-varying transverse axis of the working layout, so all
// global accesses of a wave are contiguous runs.  The forward pass parks z_i in
// the line's own unknowns of e (they are overwritten anyway and no
// concurrently processed line reads them), so no extra workspace is needed.
//
// Ordering (`mode`): 0 = one colour of the 4-colouring (jP, jQ parities),
// 1 = one hyperplane jP + 2 jQ = t of the lexicographic order (all lines of a
// hyperplane are independent and all their lexicographic predecessors lie on
// earlier hyperplanes), which reproduces the reference's
// sequential sweep.
#pragma once
#include "common.hpp"

#define EMG_LINE_BLOCK 64

template <class T>
struct LineArgs {
    int L, P, Q;
    i64 nC[3];
    FieldLayout fl;
    CellLayout cl;
    T* e;
    const T* s;
    const T* eta[3];
    const double* zeta;
    const double* h[3];
    const double* ih[3];   // 1/h
    int split;             // sweep working copies: P axis parity-split (see psplit)
    i64 mid;               // middle block of the two-sided factorisation (nL-1: one-sided)
    int xcd;               // XCD-aware workgroup -> line map
    int tile;              // lab build: switches of in-kernel instrumentation (EMG3D_Q_TILE; 256: timestamps of k_line_sweep_tha); 0 in the product
    // Everything of the above that the quad-per-block kernel needs, resolved for the axis triple (L, P, Q) on
    // the host: indexing kernel arguments with the runtime values L, P, Q costs a second, dependent
    // scalar-load round trip in the prologue of a kernel that lives for 5 us.
    struct Resolved {
        const double *ihL, *ihP, *ihQ;          // 1/h along L, P, Q
        const double *hL, *hP, *hQ;             // h along L, P, Q
        unsigned nL, csL, csP, csQ;             // blocks per line; cell strides
        unsigned nP, nQ;                        // cells along P, Q
        unsigned slot0;                         // colour mode: factor slot of the colour's first line
        unsigned off[3], st[3][3];              // field offsets / strides: component and axis in (L, P, Q) order
    } rs;
    Batch bt;              // batched systems: e, s are [system][nE]
    int qm;                // 2: mirrored two-sided factorisation (k_line_factor_m, factor_m.hpp) for k_line_sweep_thm: mid = its middle block
    int qpl;               // quad-per-block scan kernel (smooth_qpl.hpp): waves per workgroup, 0 = lane-group kernels
    int qM, seg;           // ... blocks per quad, quads per line; factor layout [line][entry][qM * seg block slots]
                           // instead of [block][entry][line]
    const unsigned char* sflag;   // level 0: [system][line slot], 1 = the line has a source entry that is not +0
                           // (k_source_line_flags); nullptr: unknown, the kernels read the source
    int zsep;              // zeta[i,j,k] == (hx_i hy_j) hz_k bit for bit (no mu_r; level 0): the sweep kernels may form it from h
    int qlpw;              // k_line_sweep_qc: lines per wave of THIS launch when > 0 (<= the instantiation's LPW; MG::q_balanced_lpw)
    int tha;               // > 0: k_line_sweep_tha (smooth_tha.hpp, affine recurrences) serves, with this many helper waves per half
    int fcomp;             // compact factor (k_line_sweep_qc, smooth_qc.hpp): 11 numbers per block, [block][entry][line]:
                           // the 4 x 4 trailing block G = W[1..4][1..4] (10) and r = 1 / S_00; W[.][0] is rebuilt in the sweep
    T* fac;
    // k_line_sweep_qpl<., 1, M> in the colour order: per-lane descriptors of THIS colour launch (element offsets of every load, the
    // coefficient products of the right-hand side and of the coupling block: everything of the prologue that depends on grid and
    // model only), written once by the kernel's generating mode; [item][qdn threads]; nullptr: the kernel computes them
    const void* qd;
    unsigned qdn;
    i64 nLinesTot;
    i64 base[4];   // first slot of colour c = cP + 2 cQ
    i64 nA[2];     // lines per colour row: number of jP with parity cP
    i64 nB2[2];    // number of jQ with parity cQ (k_line_factor, all colours in one launch)
    int mode;
    int cP, cQ;
    i64 cntA, cntB;       // mode 0
    i64 t, jQ0, cnt;      // mode 1: jQ = jQ0 + idx, jP = t - 2 jQ
                          // mode 2 (k_line_sweep_qpl only): hyperplanes t .. jQ0 in ONE workgroup, cnt = 1: descending
};

template <class T>
__device__ __forceinline__ bool line_of_thread(const LineArgs<T>& a, i64& jP, i64& jQ) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.mode == 0) {
        if (idx >= a.cntA * a.cntB) return false;
        const i64 b = idx / a.cntA, q = idx - b * a.cntA;
        jP = 1 + a.cP + 2 * q;
        jQ = 1 + a.cQ + 2 * b;
    } else {
        if (idx >= a.cnt) return false;
        jQ = a.jQ0 + idx;
        jP = a.t - 2 * jQ;
    }
    return true;
}

template <class T>
__device__ __forceinline__ i64 line_slot(const LineArgs<T>& a, i64 jP, i64 jQ) {
    const int cP = (int)((jP - 1) & 1), cQ = (int)((jQ - 1) & 1);
    return a.base[cP + 2 * cQ] + ((jP - 1) >> 1) + a.nA[cP] * ((jQ - 1) >> 1);
}

// zeta-derived coefficients of one block (reference names m{a}{b}{L|R}{c}{m|p},
// core.py:609-632, in terms of the axis triple (L,P,Q)).  S index: 0 = L, 1 = R.
struct LineCoef {
    double QP_Lm[2], PQ_Lm[2];             // at c = L, side m
    double QL_Pm[2], LQ_Pm[2], QL_Pp[2], LQ_Pp[2];
    double PL_Qm[2], LP_Qm[2], PL_Qp[2], LP_Qp[2];
};

// z[sL][sP][sQ]: zeta at cell (L: 0 = iLm, 1 = iL(clamped); P: 0 = jP-1, 1 = jP; Q alike)
__device__ __forceinline__ void line_coef(LineCoef& c, const double z[2][2][2], const double kL[2],
                                          const double kP[2], const double kQ[2]) {
#pragma unroll
    for (int S = 0; S < 2; ++S) {
        c.QP_Lm[S] = kP[S] * (z[0][S][1] + z[0][S][0]);
        c.PQ_Lm[S] = kQ[S] * (z[0][1][S] + z[0][0][S]);
        c.QL_Pm[S] = kL[S] * (z[S][0][1] + z[S][0][0]);
        c.LQ_Pm[S] = kQ[S] * (z[1][0][S] + z[0][0][S]);
        c.QL_Pp[S] = kL[S] * (z[S][1][1] + z[S][1][0]);
        c.LQ_Pp[S] = kQ[S] * (z[1][1][S] + z[0][1][S]);
        c.PL_Qm[S] = kL[S] * (z[S][1][0] + z[S][0][0]);
        c.LP_Qm[S] = kP[S] * (z[1][S][0] + z[0][S][0]);
        c.PL_Qp[S] = kL[S] * (z[S][1][1] + z[S][0][1]);
        c.LP_Qp[S] = kP[S] * (z[1][S][1] + z[0][S][1]);
    }
}

// Sub-diagonal block A (`left`, core.py:684-691): row 0 = a[1..4], diag = d[1..4].
__device__ __forceinline__ void line_left(const LineCoef& c, double ihLm, double a[5], double d[5]) {
    a[0] = 0.0; d[0] = 0.0;
    a[1] = c.QP_Lm[0] * ihLm;
    a[2] = -c.QP_Lm[1] * ihLm;
    a[3] = c.PQ_Lm[0] * ihLm;
    a[4] = -c.PQ_Lm[1] * ihLm;
    d[1] = -c.QL_Pm[0] * ihLm;
    d[2] = -c.QL_Pp[0] * ihLm;
    d[3] = -c.PL_Qm[0] * ihLm;
    d[4] = -c.PL_Qp[0] * ihLm;
}

// Packed index of the symmetric block inverse W = S^{-1}: row-major lower
// triangle, (r, c) with r >= c  ->  r (r + 1) / 2 + c   (15 numbers per block).
__host__ __device__ __forceinline__ constexpr int wpk(int r, int c) {
    return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r;
}

// y <- W y  (explicit symmetric 5x5 inverse).
template <class T>
__device__ __forceinline__ void apply_sinv(const T f[15], T y[5]) {
    T o[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        T t = f[wpk(r, 0)] * y[0];
#pragma unroll
        for (int c = 1; c < 5; ++c) t += f[wpk(r, c)] * y[c];
        o[r] = t;
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) y[r] = o[r];
}

// ---------------------------------------------------------------------------
// Factorisation kernel: thread per line.  Two-sided ("twisted") elimination
// around the middle block a.mid:
//   left chain   i = 0 .. mid-1   : S_i = M_i - A_i W_{i-1} A_i^T
//   right chain  i = nL-1 .. mid+1: S_i = M_i - A_{i+1}^T W_{i+1} A_{i+1}
//   middle       i = mid          : S_m = M_m - A_m W_{m-1} A_m^T - A_{m+1}^T W_{m+1} A_{m+1}
// with W_i = S_i^{-1} stored for every block.  mid = nL-1 gives the plain
// one-sided band LDL^T of the reference (core.py:1447-1582) in block form;
// mid = (nL-1)/2 halves the length of the recurrences in the sweep.
// ---------------------------------------------------------------------------
template <class T>
struct BlockMat {
    T S[5][5];          // lower triangle of the 5x5 `middle` block (core.py:653-681)
    double al[5], dl[5];  // `left` block: row 0 (al[1..4]) and diagonal (dl[1..4]) (core.py:684-691)
};

template <class T>
__device__ __forceinline__ void line_block(const LineArgs<T>& a, i64 i, i64 jP, i64 jQ, BlockMat<T>& bm) {
    const int L = a.L, P = a.P, Q = a.Q;
    const i64 nL = a.nC[L];
    const i64 iL = (i + 1 < nL) ? i + 1 : nL - 1;   // clamp, core.py:605
    const i64 csL = a.cl.st[L], csP = a.cl.st[P], csQ = a.cl.st[Q];
    const i64 cbase = (jP - 1) * csP + (jQ - 1) * csQ;
    const double hP[2] = {a.h[P][jP - 1], a.h[P][jP]};
    const double hQ[2] = {a.h[Q][jQ - 1], a.h[Q][jQ]};
    const double kP[2] = {0.5 / hP[0], 0.5 / hP[1]};
    const double kQ[2] = {0.5 / hQ[0], 0.5 / hQ[1]};
    const double kL[2] = {0.5 / a.h[L][i], 0.5 / a.h[L][iL]};
    const double ihL[2] = {1.0 / a.h[L][i], 1.0 / a.h[L][iL]};
    double z[2][2][2];
    T etL[2][2], etP[2][2][2], etQ[2][2][2];
#pragma unroll
    for (int sP = 0; sP < 2; ++sP)
#pragma unroll
        for (int sQ = 0; sQ < 2; ++sQ) {
            const i64 p0 = cbase + i * csL + sP * csP + sQ * csQ;
            const i64 p1 = cbase + iL * csL + sP * csP + sQ * csQ;
            z[0][sP][sQ] = a.zeta[p0]; z[1][sP][sQ] = a.zeta[p1];
            etL[sP][sQ] = a.eta[L][p0];
            etP[0][sP][sQ] = a.eta[P][p0]; etP[1][sP][sQ] = a.eta[P][p1];
            etQ[0][sP][sQ] = a.eta[Q][p0]; etQ[1][sP][sQ] = a.eta[Q][p1];
        }
    LineCoef c;
    line_coef(c, z, kL, kP, kQ);
    T st[5];   // eta sums (core.py:635-648)
    st[0] = ((etL[1][1] + etL[1][0]) + etL[0][1]) + etL[0][0];
    st[1] = ((etP[1][0][1] + etP[1][0][0]) + etP[0][0][1]) + etP[0][0][0];
    st[2] = ((etP[1][1][1] + etP[1][1][0]) + etP[0][1][1]) + etP[0][1][0];
    st[3] = ((etQ[1][1][0] + etQ[1][0][0]) + etQ[0][1][0]) + etQ[0][0][0];
    st[4] = ((etQ[1][1][1] + etQ[1][0][1]) + etQ[0][1][1]) + etQ[0][0][1];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) bm.S[r][cc] = Zero<T>::v();
#pragma unroll
    for (int q = 0; q < 5; ++q) bm.S[q][q] = -(st[q] * 0.25);
    add_real(bm.S[0][0], c.QP_Lm[1] / hP[1] + c.QP_Lm[0] / hP[0]);
    add_real(bm.S[0][0], c.PQ_Lm[1] / hQ[1] + c.PQ_Lm[0] / hQ[0]);
    add_real(bm.S[1][1], c.QL_Pm[1] * ihL[1] + c.QL_Pm[0] * ihL[0]);
    add_real(bm.S[1][1], c.LQ_Pm[1] / hQ[1] + c.LQ_Pm[0] / hQ[0]);
    add_real(bm.S[2][2], c.QL_Pp[1] * ihL[1] + c.QL_Pp[0] * ihL[0]);
    add_real(bm.S[2][2], c.LQ_Pp[1] / hQ[1] + c.LQ_Pp[0] / hQ[0]);
    add_real(bm.S[3][3], c.PL_Qm[1] * ihL[1] + c.PL_Qm[0] * ihL[0]);
    add_real(bm.S[3][3], c.LP_Qm[1] / hP[1] + c.LP_Qm[0] / hP[0]);
    add_real(bm.S[4][4], c.PL_Qp[1] * ihL[1] + c.PL_Qp[0] * ihL[0]);
    add_real(bm.S[4][4], c.LP_Qp[1] / hP[1] + c.LP_Qp[0] / hP[0]);
    add_real(bm.S[1][0], -c.QP_Lm[0] * ihL[0]);
    add_real(bm.S[2][0], c.QP_Lm[1] * ihL[0]);
    add_real(bm.S[3][0], -c.PQ_Lm[0] * ihL[0]);
    add_real(bm.S[4][0], c.PQ_Lm[1] * ihL[0]);
    add_real(bm.S[3][1], -c.LQ_Pm[0] / hP[0]);
    add_real(bm.S[4][1], c.LQ_Pm[1] / hP[0]);
    add_real(bm.S[3][2], c.LQ_Pp[0] / hP[1]);
    add_real(bm.S[4][2], -c.LQ_Pp[1] / hP[1]);
    line_left(c, ihL[0], bm.al, bm.dl);
    if (i == nL - 1) {     // last block: only row 0 of `left` exists (core.py:1434-1444)
#pragma unroll
        for (int q = 0; q < 5; ++q) bm.dl[q] = 0.0;
    }
}

// S -= A W A^T (A = `left` of THIS block; only W[1..4][1..4] enters).  only00: 1x1 block.
template <class T>
__device__ __forceinline__ void schur_left(T S[5][5], const double al[5], const double dl[5],
                                           const T W[5][5], bool only00) {
    T Wa[5];
#pragma unroll
    for (int r = 1; r < 5; ++r) {
        T t = Zero<T>::v();
#pragma unroll
        for (int q = 1; q < 5; ++q) t += W[r][q] * al[q];
        Wa[r] = t;
    }
    T aWa = Zero<T>::v();
#pragma unroll
    for (int r = 1; r < 5; ++r) aWa += Wa[r] * al[r];
    S[0][0] -= aWa;
    if (!only00) {
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            S[r][0] -= Wa[r] * dl[r];
#pragma unroll
            for (int cc = 1; cc <= r; ++cc) S[r][cc] -= W[r][cc] * (dl[r] * dl[cc]);
        }
    }
}

// S -= A^T W A (A = `left` of the NEXT block, W its inverse): changes rows/cols 1..4 only.
template <class T>
__device__ __forceinline__ void schur_right(T S[5][5], const double al[5], const double dl[5],
                                            const T W[5][5]) {
#pragma unroll
    for (int r = 1; r < 5; ++r)
#pragma unroll
        for (int cc = 1; cc <= r; ++cc) {
            T t = W[0][0] * (al[r] * al[cc]);
            t += W[0][cc] * (al[r] * dl[cc]);
            t += W[r][0] * (dl[r] * al[cc]);
            t += W[r][cc] * (dl[r] * dl[cc]);
            S[r][cc] -= t;
        }
}

// W = S^{-1} (symmetric, via non-pivoting LDL^T as core.solve); only00: 1x1 block.
template <class T>
__device__ __forceinline__ void invert_block(const T S[5][5], T W[5][5], bool only00) {
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) W[r][cc] = Zero<T>::v();
    if (only00) { W[0][0] = recip(S[0][0]); return; }
    T D[5], Dinv[5], Lm[5][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        T dj = S[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= (Lm[j][k] * Lm[j][k]) * D[k];
        D[j] = dj;
        const T inv = recip(dj);
        Dinv[j] = inv;
#pragma unroll
        for (int r = j + 1; r < 5; ++r) {
            T v = S[r][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= (Lm[r][k] * Lm[j][k]) * D[k];
            Lm[r][j] = v * inv;
        }
    }
    T N[5][5];   // N = L^{-1} (unit lower)
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int r = c + 1; r < 5; ++r) {
            T t = -Lm[r][c];
#pragma unroll
            for (int k = c + 1; k < r; ++k) t -= Lm[r][k] * N[k][c];
            N[r][c] = t;
        }
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int cc = 0; cc <= r; ++cc) {
            T t = Zero<T>::v();   // sum over m >= r of N[m][r] Dinv[m] N[m][cc]  (N[m][m] = 1)
#pragma unroll
            for (int m = r; m < 5; ++m) {
                const T nr = (m == r) ? Dinv[m] : N[m][r] * Dinv[m];
                t += (m == cc) ? nr : nr * N[m][cc];
            }
            W[r][cc] = t;
            W[cc][r] = t;
        }
}

// packed index of G = W[1..4][1..4] (row-major lower triangle of the 4 x 4 block), entry 10 = r
__host__ __device__ __forceinline__ constexpr int gpk(int k, int j) {     // k, j in 1..4
    return k >= j ? (k - 1) * k / 2 + (j - 1) : (j - 1) * j / 2 + (k - 1);
}
template <class T>
__device__ __forceinline__ void store_block(const LineArgs<T>& a, i64 i, i64 slot, const T W[5][5], T r00 = Zero<T>::v()) {
    if (a.fcomp) {
        T* dst = a.fac + (i * 11) * a.nLinesTot + slot;
#pragma unroll
        for (int k = 1; k < 5; ++k)
#pragma unroll
            for (int j = 1; j <= k; ++j) dst[(i64)gpk(k, j) * a.nLinesTot] = W[k][j];
        dst[(i64)10 * a.nLinesTot] = r00;
        return;
    }
    if (a.qpl) {
        const i64 per = (i64)a.qM * a.seg;
        T* dst = a.fac + slot * 15 * per + i;
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) dst[(i64)wpk(r, cc) * per] = W[r][cc];
        return;
    }
    T* dst = a.fac + (i * 15) * a.nLinesTot + slot;   // [block][entry][line]
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int cc = 0; cc <= r; ++cc) dst[(i64)wpk(r, cc) * a.nLinesTot] = W[r][cc];
}

// Which lines carry a source?  One thread per line of (level, direction), all four colours (blockIdx.y): flags[slot] = 1 when
// any of the line's 5 nL - 4 source entries differs from +0 IN ITS BITS (-0 counts as a value: skipping it could flip the
// sign of a zero sum).  `s`, `fl`: the source in the reference layout.  The dipole source of a survey marks a handful of
// lines; the sweep kernels skip the source loads of waves whose lines are all clear (smooth_qc.hpp).
__device__ __forceinline__ bool bits_nonzero(double v) { return __double_as_longlong(v) != 0; }
__device__ __forceinline__ bool bits_nonzero(c128 v) { return (__double_as_longlong(v.re) | __double_as_longlong(v.im)) != 0; }
template <class T>
__global__ __launch_bounds__(EMG_LINE_BLOCK) void k_source_line_flags(LineArgs<T> a, const T* __restrict__ s, FieldLayout fl,
                                                                      unsigned char* __restrict__ flags, i64 sys_stride) {
    s += (i64)blockIdx.z * sys_stride;                  // batched systems: blockIdx.z = system (frozen ones included)
    flags += (i64)blockIdx.z * a.nLinesTot;
    const int cP = blockIdx.y & 1, cQ = blockIdx.y >> 1;
    const i64 cntA = a.nA[cP], idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cntA * a.nB2[cQ]) return;
    const i64 b = idx / cntA, q = idx - b * cntA;
    const i64 jP = 1 + cP + 2 * q, jQ = 1 + cQ + 2 * b;
    const int L = a.L, P = a.P, Q = a.Q;
    const i64 nL = a.nC[L];
    bool any = false;
    for (i64 i = 0; i < nL; ++i) {
        any = any || bits_nonzero(s[fl.off[L] + i * fl.st[L][L] + jP * fl.st[L][P] + jQ * fl.st[L][Q]]);
        if (i + 1 < nL) {
            const i64 pb = fl.off[P] + (i + 1) * fl.st[P][L] + jQ * fl.st[P][Q];
            const i64 qb = fl.off[Q] + (i + 1) * fl.st[Q][L] + jP * fl.st[Q][P];
            any = any || bits_nonzero(s[pb + (jP - 1) * fl.st[P][P]]) || bits_nonzero(s[pb + jP * fl.st[P][P]]) ||
                  bits_nonzero(s[qb + (jQ - 1) * fl.st[Q][Q]]) || bits_nonzero(s[qb + jQ * fl.st[Q][Q]]);
        }
    }
    flags[line_slot(a, jP, jQ)] = any ? 1 : 0;
}

template <class T>
__global__ __launch_bounds__(EMG_LINE_BLOCK) void k_line_factor(LineArgs<T> a) {
    i64 jP, jQ;
    if (a.mode == 3) {      // all four colours in one launch: blockIdx.y = colour (the lines are independent)
        const int cP = blockIdx.y & 1, cQ = blockIdx.y >> 1;
        const i64 cntA = a.nA[cP], idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= cntA * a.nB2[cQ]) return;
        const i64 b = idx / cntA, q = idx - b * cntA;
        jP = 1 + cP + 2 * q;
        jQ = 1 + cQ + 2 * b;
    } else if (!line_of_thread(a, jP, jQ)) return;
    const i64 nL = a.nC[a.L];
    const i64 mid = a.mid;
    const i64 slot = line_slot(a, jP, jQ);
    BlockMat<T> bm;
    T W[5][5], WL[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) { W[r][c] = Zero<T>::v(); WL[r][c] = Zero<T>::v(); }
    // left chain
    for (i64 i = 0; i < mid; ++i) {
        line_block(a, i, jP, jQ, bm);
        if (i > 0) schur_left(bm.S, bm.al, bm.dl, W, false);
        const T r00 = a.fcomp ? recip(bm.S[0][0]) : Zero<T>::v();      // (= the first pivot's reciprocal of invert_block)
        invert_block(bm.S, W, false);
        store_block(a, i, slot, W, r00);
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) { WL[r][c] = W[r][c]; W[r][c] = Zero<T>::v(); }
    // right chain
    double alN[5] = {0, 0, 0, 0, 0}, dlN[5] = {0, 0, 0, 0, 0};
    for (i64 i = nL - 1; i > mid; --i) {
        line_block(a, i, jP, jQ, bm);
        const bool lastb = (i == nL - 1);
        if (!lastb) schur_right(bm.S, alN, dlN, W);
        invert_block(bm.S, W, lastb);
        store_block(a, i, slot, W);
#pragma unroll
        for (int q = 0; q < 5; ++q) { alN[q] = bm.al[q]; dlN[q] = bm.dl[q]; }
    }
    // middle block
    {
        line_block(a, mid, jP, jQ, bm);
        const bool lastb = (mid == nL - 1);
        if (mid > 0) schur_left(bm.S, bm.al, bm.dl, WL, lastb);
        if (!lastb) schur_right(bm.S, alN, dlN, W);
        const T r00 = a.fcomp ? recip(bm.S[0][0]) : Zero<T>::v();
        invert_block(bm.S, W, lastb);
        store_block(a, mid, slot, W, r00);
    }
}

// ---------------------------------------------------------------------------
// Sweep kernel: forward + backward substitution of one set of independent
// lines (one colour or one hyperplane).
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(EMG_LINE_BLOCK) void k_line_sweep(LineArgs<T> a) {
    EMG_BATCH(y, a.bt);
    i64 jP, jQ;
    if (!line_of_thread(a, jP, jQ)) return;
    const int L = a.L, P = a.P, Q = a.Q;
    const i64 nL = a.nC[L];
    const i64 slot = line_slot(a, jP, jQ);
    const i64 csL = a.cl.st[L], csP = a.cl.st[P], csQ = a.cl.st[Q];
    const double* hL = a.h[L];
    const double hP[2] = {a.h[P][jP - 1], a.h[P][jP]};
    const double hQ[2] = {a.h[Q][jQ - 1], a.h[Q][jQ]};
    const double kP[2] = {0.5 / hP[0], 0.5 / hP[1]};
    const double kQ[2] = {0.5 / hQ[0], 0.5 / hQ[1]};
    const double ihP[2] = {1.0 / hP[0], 1.0 / hP[1]};
    const double ihQ[2] = {1.0 / hQ[0], 1.0 / hQ[1]};
    const i64 cbase = (jP - 1) * csP + (jQ - 1) * csQ;

    // field addressing: component c at (vL, vP, vQ)
    const FieldLayout& fl = a.fl;
    const i64 oL = fl.off[L], sLL = fl.st[L][L], sLP = fl.st[L][P], sLQ = fl.st[L][Q];
    const i64 oP = fl.off[P], sPL = fl.st[P][L], sPP = fl.st[P][P], sPQ = fl.st[P][Q];
    const i64 oQ = fl.off[Q], sQL = fl.st[Q][L], sQP = fl.st[Q][P], sQQ = fl.st[Q][Q];
#define FL_(vL, vP, vQ) (oL + (vL) * sLL + (vP) * sLP + (vQ) * sLQ)
#define FP_(vL, vP, vQ) (oP + (vL) * sPL + (vP) * sPP + (vQ) * sPQ)
#define FQ_(vL, vP, vQ) (oQ + (vL) * sQL + (vP) * sQP + (vQ) * sQQ)
    T* e = (a.e + boff_);
    const T* s = (a.s + boff_);
    const i64 jPm = jP - 1, jPp = jP + 1, jQm = jQ - 1, jQp = jQ + 1;

    double z[2][2][2];
#pragma unroll
    for (int sP = 0; sP < 2; ++sP)
#pragma unroll
        for (int sQ = 0; sQ < 2; ++sQ) z[1][sP][sQ] = a.zeta[cbase + sP * csP + sQ * csQ];

    T zprev[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) zprev[q] = Zero<T>::v();

    // ----------------------------- forward ---------------------------------
    for (i64 i = 0; i < nL; ++i) {
        const i64 iLm = i;
        const i64 iL = (i + 1 < nL) ? i + 1 : nL - 1;
        const bool last = (i == nL - 1);
#pragma unroll
        for (int sP = 0; sP < 2; ++sP)
#pragma unroll
            for (int sQ = 0; sQ < 2; ++sQ) {
                z[0][sP][sQ] = z[1][sP][sQ];
                if (!last) z[1][sP][sQ] = a.zeta[cbase + iL * csL + sP * csP + sQ * csQ];
            }
        const double kL[2] = {0.5 / hL[iLm], 0.5 / hL[iL]};
        const double ihLm = 1.0 / hL[iLm];
        LineCoef c;
        line_coef(c, z, kL, kP, kQ);

        // factor of this block
        T f[15];
        const T* src = a.fac + (i * 15) * a.nLinesTot + slot;
#pragma unroll
        for (int q = 0; q < 15; ++q) f[q] = src[q * a.nLinesTot];

        // right-hand side (core.py:697-736)
        T y[5];
        y[0] = s[FL_(iLm, jP, jQ)];
        y[0] += (c.QP_Lm[1] * e[FL_(iLm, jPp, jQ)]) * ihP[1];
        y[0] += (c.QP_Lm[0] * e[FL_(iLm, jPm, jQ)]) * ihP[0];
        y[0] += (c.PQ_Lm[1] * e[FL_(iLm, jP, jQp)]) * ihQ[1];
        y[0] += (c.PQ_Lm[0] * e[FL_(iLm, jP, jQm)]) * ihQ[0];
        if (!last) {
            y[1] = s[FP_(iL, jPm, jQ)];
            y[2] = s[FP_(iL, jP, jQ)];
            y[3] = s[FQ_(iL, jP, jQm)];
            y[4] = s[FQ_(iL, jP, jQ)];

            y[1] += (c.QL_Pm[1] * e[FL_(iL, jPm, jQ)] - c.QL_Pm[0] * e[FL_(iLm, jPm, jQ)] +
                     c.LQ_Pm[1] * e[FQ_(iL, jPm, jQ)] - c.LQ_Pm[0] * e[FQ_(iL, jPm, jQm)]) * ihP[0];
            y[1] += (c.LQ_Pm[1] * e[FP_(iL, jPm, jQp)]) * ihQ[1];
            y[1] += (c.LQ_Pm[0] * e[FP_(iL, jPm, jQm)]) * ihQ[0];

            y[2] += (c.QL_Pp[0] * e[FL_(iLm, jPp, jQ)] - c.QL_Pp[1] * e[FL_(iL, jPp, jQ)] +
                     c.LQ_Pp[0] * e[FQ_(iL, jPp, jQm)] - c.LQ_Pp[1] * e[FQ_(iL, jPp, jQ)]) * ihP[1];
            y[2] += (c.LQ_Pp[1] * e[FP_(iL, jP, jQp)]) * ihQ[1];
            y[2] += (c.LQ_Pp[0] * e[FP_(iL, jP, jQm)]) * ihQ[0];

            y[3] += (c.PL_Qm[1] * e[FL_(iL, jP, jQm)] - c.PL_Qm[0] * e[FL_(iLm, jP, jQm)] +
                     c.LP_Qm[1] * e[FP_(iL, jP, jQm)] - c.LP_Qm[0] * e[FP_(iL, jPm, jQm)]) * ihQ[0];
            y[3] += (c.LP_Qm[1] * e[FQ_(iL, jPp, jQm)]) * ihP[1];
            y[3] += (c.LP_Qm[0] * e[FQ_(iL, jPm, jQm)]) * ihP[0];

            y[4] += (c.PL_Qp[0] * e[FL_(iLm, jP, jQp)] - c.PL_Qp[1] * e[FL_(iL, jP, jQp)] +
                     c.LP_Qp[0] * e[FP_(iL, jPm, jQp)] - c.LP_Qp[1] * e[FP_(iL, jP, jQp)]) * ihQ[1];
            y[4] += (c.LP_Qp[1] * e[FQ_(iL, jPp, jQ)]) * ihP[1];
            y[4] += (c.LP_Qp[0] * e[FQ_(iL, jPm, jQ)]) * ihP[0];
        } else {
#pragma unroll
            for (int q = 1; q < 5; ++q) y[q] = Zero<T>::v();
        }

        // y -= A z_{i-1}
        if (i > 0) {
            double al[5], dl[5];
            line_left(c, ihLm, al, dl);
            T t0 = Zero<T>::v();
#pragma unroll
            for (int q = 1; q < 5; ++q) t0 += zprev[q] * al[q];
            y[0] -= t0;
            if (!last) {
#pragma unroll
                for (int q = 1; q < 5; ++q) y[q] -= zprev[q] * dl[q];
            }
        }
        apply_sinv(f, y);
#pragma unroll
        for (int q = 0; q < 5; ++q) zprev[q] = y[q];
        // park z_i in the line's own unknowns
        e[FL_(iLm, jP, jQ)] = y[0];
        if (!last) {
            e[FP_(iL, jPm, jQ)] = y[1];
            e[FP_(iL, jP, jQ)] = y[2];
            e[FQ_(iL, jP, jQm)] = y[3];
            e[FQ_(iL, jP, jQ)] = y[4];
        }
    }

    // ----------------------------- backward --------------------------------
    // x_last = z_last (already stored).  zprev holds x_{i+1}.
    // zeta window: z[0] currently holds cells at L-index nL-1.
    for (i64 i = nL - 2; i >= 0; --i) {
        // coefficients of block i+1 (its iLm = i+1): need cells at L-index i+1
        // (side 0 of that block) -> only the c=L,sc=0 and b=L,sb=0 coefficients.
        const i64 iN = i + 1;
        double zc[2][2];
#pragma unroll
        for (int sP = 0; sP < 2; ++sP)
#pragma unroll
            for (int sQ = 0; sQ < 2; ++sQ) zc[sP][sQ] = a.zeta[cbase + iN * csL + sP * csP + sQ * csQ];
        const double kLn = 0.5 / hL[iN], ihLn = 1.0 / hL[iN];
        double al[5], dl[5];
        al[1] = (kP[0] * (zc[0][1] + zc[0][0])) * ihLn;
        al[2] = -(kP[1] * (zc[1][1] + zc[1][0])) * ihLn;
        al[3] = (kQ[0] * (zc[1][0] + zc[0][0])) * ihLn;
        al[4] = -(kQ[1] * (zc[1][1] + zc[0][1])) * ihLn;
        dl[1] = -(kLn * (zc[0][1] + zc[0][0])) * ihLn;
        dl[2] = -(kLn * (zc[1][1] + zc[1][0])) * ihLn;
        dl[3] = -(kLn * (zc[1][0] + zc[0][0])) * ihLn;
        dl[4] = -(kLn * (zc[1][1] + zc[0][1])) * ihLn;
        const bool nextlast = (iN == nL - 1);

        T f[15];
        const T* src = a.fac + (i * 15) * a.nLinesTot + slot;
#pragma unroll
        for (int q = 0; q < 15; ++q) f[q] = src[q * a.nLinesTot];

        // v = A_{i+1}^T x_{i+1}: v_0 = 0, v_k = a_k x0 + d_k x_k
        T v[5];
        v[0] = Zero<T>::v();
#pragma unroll
        for (int q = 1; q < 5; ++q) {
            v[q] = zprev[0] * al[q];
            if (!nextlast) v[q] += zprev[q] * dl[q];
        }
        apply_sinv(f, v);
        const i64 iL = i + 1;
        T x[5];
        x[0] = e[FL_(i, jP, jQ)] - v[0];
        x[1] = e[FP_(iL, jPm, jQ)] - v[1];
        x[2] = e[FP_(iL, jP, jQ)] - v[2];
        x[3] = e[FQ_(iL, jP, jQm)] - v[3];
        x[4] = e[FQ_(iL, jP, jQ)] - v[4];
        e[FL_(i, jP, jQ)] = x[0];
        e[FP_(iL, jPm, jQ)] = x[1];
        e[FP_(iL, jP, jQ)] = x[2];
        e[FQ_(iL, jP, jQm)] = x[3];
        e[FQ_(iL, jP, jQ)] = x[4];
#pragma unroll
        for (int q = 0; q < 5; ++q) zprev[q] = x[q];
    }
#undef FL_
#undef FP_
#undef FQ_
}

// ---------------------------------------------------------------------------
// Row-parallel sweep kernel (the production line smoother).
//
// One line is handled by a group of 8 consecutive lanes; lane r < 5 owns row r
// of every 5x5 block (r = 0: the edge along the line; r = 1,2: the two
// P-directed edges at node i+1; r = 3,4: the two Q-directed edges), lanes 5-7
// idle.  A wave therefore advances 8 lines at once and a grid has 8x the waves
// of a thread-per-line launch -- the recurrence along the line is latency
// bound, so wave count is what buys throughput.  Per block step a lane
//   * evaluates ITS row of the right-hand side: s + sum_t g_t E_t with six
//     neighbour values E_t and coefficients g_t built from a 2x2 face of zeta
//     (the reference's m-coefficients, core.py:609-632, regrouped per row),
//   * forms its coupling to the previous block (row 0 needs a sum over lanes
//     1..4: one 8-lane butterfly), gathers the five y values of the group and
//     multiplies with its row of the cached symmetric inverse W_i.
// All loads of step i+1 are issued before the arithmetic of step i (software
// prefetch in registers), which hides the HBM/L2 latency of the dependent chain.
// ---------------------------------------------------------------------------
#ifndef EMG_RP_BLOCK
#define EMG_RP_BLOCK 256
#endif

// EMG_LPW lines per wave: row r of line g lives in lane EMG_LPW*r + g (r < 5);
// the remaining 64 - 5*EMG_LPW lanes mirror row 0 of the first lines (no stores).
// Tuned on MI355X at 128^3 (A/B in one session): 4 lines per wave and a
// 3-deep register prefetch are ~10 % faster than 8 lines / 2-deep; 12 lines per
// wave or 1-2 lines per wave are 25-200 % slower.
#ifndef EMG_LPW
#define EMG_LPW 4
#endif
#ifndef EMG_RP_STAGES
#define EMG_RP_STAGES 3
#endif
template <class T> __device__ __forceinline__ T shfl_row(T v, int src, int g);
template <> __device__ __forceinline__ double shfl_row<double>(double v, int src, int g) {
    return __shfl(v, EMG_LPW * src + g, 64);
}
template <> __device__ __forceinline__ c128 shfl_row<c128>(c128 v, int src, int g) {
    const int l = EMG_LPW * src + g;
    return mk(__shfl(v.re, l, 64), __shfl(v.im, l, 64));
}

template <class T>
struct RpStep {       // everything a lane loads for one forward block step
    T W[5];
    T E[6];
    T S;
    double zf[4];     // zeta face: [u][v] -> zf[2u+v]
    double ihl0, ihl1;  // 1/hL[i], 1/hL[iL]
};

template <class T>
struct RpBack {       // ... and for one backward step
    T W[5];
    T zi;
    double p0, p1, ihln;
};

template <class T, int LPW>
__global__ __launch_bounds__(EMG_RP_BLOCK) void k_line_sweep_rp(LineArgs<T> a) {
    // lane = LPW r + g: the lanes that hold the SAME row of consecutive lines
    // are adjacent, so a quad of lanes reads neighbouring addresses (the
    // address unit coalesces per quad); the rows of one line sit LPW lanes apart.
    const int lane = threadIdx.x & 63;
    const int r = lane / LPW;                 // >= 5: mirror lanes
    const int g = lane - r * LPW;
    // XCD-aware: workgroup b runs on XCD b % 8 and takes the (b % 8)-th eighth of the line slots, so
    // that lines which share neighbour values (adjacent in Q) meet in the same L2
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
    const i64 nL = a.nC[L];
    const i64 slot = line_slot(a, jP, jQ);
    const i64 nLt = a.nLinesTot;
    const i64 csL = a.cl.st[L], csP = a.cl.st[P], csQ = a.cl.st[Q];
    const double ihP[2] = {a.ih[P][jP - 1], a.ih[P][jP]};
    const double ihQ[2] = {a.ih[Q][jQ - 1], a.ih[Q][jQ]};
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    const FieldLayout& fl = a.fl;
    const i64 jPm = jP - 1, jPp = jP + 1, jQm = jQ - 1, jQp = jQ + 1;
    // P coordinate -> storage position (parity split in the working copies)
    const i64 nPc = a.nC[P], nPn = a.nC[P] + 1;
    const bool spl = (a.split & 1) != 0;
#define SPC_(v) (spl ? psplit((v), nPc) : (v))
#define SPN_(v) (spl ? psplit((v), nPn) : (v))
#define FL_(vL, vP, vQ) (fl.off[L] + (vL) * fl.st[L][L] + SPN_(vP) * fl.st[L][P] + (vQ) * fl.st[L][Q])
#define FP_(vL, vP, vQ) (fl.off[P] + (vL) * fl.st[P][L] + SPC_(vP) * fl.st[P][P] + (vQ) * fl.st[P][Q])
#define FQ_(vL, vP, vQ) (fl.off[Q] + (vL) * fl.st[Q][L] + SPN_(vP) * fl.st[Q][P] + (vQ) * fl.st[Q][Q])
    const i64 cP0 = SPC_(jP - 1) * csP, cP1 = SPC_(jP) * csP, cq = (jQ - 1) * csQ;

    // ---- per-lane (row) configuration --------------------------------------
    // Lanes 5..7 mirror lane 0 (same addresses, same arithmetic, no stores) so
    // that the loop bodies are free of divergent branches: every load is
    // unconditional, which lets the compiler keep the prefetched loads in
    // flight (counted vmcnt) instead of draining them at control-flow joins.
    const bool rowact = r < 5;
    const int rr = rowact ? r : 0;
    const int type = (rr == 0) ? 0 : (rr <= 2 ? 1 : 2);   // 0: L-row, 1: P-rows, 2: Q-rows
    const int side = (rr == 0) ? 0 : ((rr - 1) & 1);      // fixed transverse side (0 minus, 1 plus)
    const double sg = side ? -1.0 : 1.0;
    const double tmask = (type == 0) ? 0.0 : 1.0;         // transverse row?
    i64 ob[7], os[7];          // offsets (s/self, E1..E6) at step 0 and per-step strides
    i64 fb, sv, suT0;          // zeta face base, v-stride, u-stride for type 0
    double K[6];               // per-lane constant factors of the six coefficients
    double ca = 0.0;           // a_k = ca * rowsum0 / hL[i]
    if (type == 0) {
        ob[0] = FL_(0, jP, jQ);
        ob[1] = FL_(0, jPp, jQ); ob[2] = FL_(0, jPm, jQ); ob[3] = FL_(0, jP, jQp); ob[4] = FL_(0, jP, jQm);
        ob[5] = ob[1]; ob[6] = ob[1];
#pragma unroll
        for (int t = 0; t < 7; ++t) os[t] = fl.st[L][L];
        fb = cP0 + cq; sv = csQ; suT0 = cP1 - cP0;
        K[0] = kP[1] * ihP[1]; K[1] = kP[0] * ihP[0]; K[2] = kQ[1] * ihQ[1]; K[3] = kQ[0] * ihQ[0];
        K[4] = 0.0; K[5] = 0.0;
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
        K[0] = sg * ihA; K[1] = -sg * ihA;                      // x kL[1], x kL[0] per step
        K[2] = sg * kQ[1] * ihA; K[3] = -sg * kQ[0] * ihA;
        K[4] = kQ[1] * ihQ[1]; K[5] = kQ[0] * ihQ[0];
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
        K[0] = sg * ihA; K[1] = -sg * ihA;
        K[2] = sg * kP[1] * ihA; K[3] = -sg * kP[0] * ihA;
        K[4] = kP[1] * ihP[1]; K[5] = kP[0] * ihP[0];
        ca = sg * 0.5 * ihA;
    }
#undef FL_
#undef FP_
#undef FQ_
#undef SPC_
#undef SPN_
    const bool t0 = (type == 0);
    const double t0f = t0 ? 1.0 : 0.0;
    const i64 wstep = 15 * nLt;

    // Addressing: uniform (scalar) base pointers that advance per block plus
    // 32-bit per-lane BYTE offsets -> `global_load v, voff, s[base]` with one
    // 32-bit VALU add per load instead of 64-bit pointer arithmetic.  The host
    // only selects this kernel when every array is < 4 GiB.
    typedef unsigned int u32;
    const char* const eB = reinterpret_cast<const char*>((a.e + boff_));
    const char* const sB = reinterpret_cast<const char*>((a.s + boff_));
    u32 wo[5];                       // per-lane offsets of its W row inside one block record
#pragma unroll
    for (int c = 0; c < 5; ++c) wo[c] = (u32)(((i64)wpk(rr, c) * nLt + slot) * (i64)sizeof(T));
    u32 eo[6], es[6];                // field offsets (advance per block by es)
#pragma unroll
    for (int t = 0; t < 6; ++t) { eo[t] = (u32)(ob[1 + t] * (i64)sizeof(T)); es[t] = (u32)(os[1 + t] * (i64)sizeof(T)); }
    u32 so = (u32)(ob[0] * (i64)sizeof(T));
    const u32 ss = (u32)(os[0] * (i64)sizeof(T));
    const u32 zo0 = (u32)(fb * 8), zo1 = (u32)((fb + sv) * 8);   // zeta face offsets (u = 0)
    const u32 zsu = (u32)(suT0 * 8);                             // type-0 u-stride (bytes, modular)
    const u32 zsL = (u32)(csL * 8);

    // Wave-private LDS exchange buffers (row c of line g sits at index LPW*c+g)
    __shared__ T xch[EMG_RP_BLOCK / 64][2][64];
    T* const xu = xch[threadIdx.x >> 6][0];
    T* const xy = xch[threadIdx.x >> 6][1];

    // ----------------------------- forward ---------------------------------
    const char* wB = reinterpret_cast<const char*>(a.fac);       // + block * wstep (uniform)
    const char* zB = reinterpret_cast<const char*>(a.zeta);      // + block * csL (uniform)
    const double* hB = a.ih[L];
    // Unconditional loads.  For the last block the transverse rows do not
    // exist: their field loads are clamped to the previous block (valid
    // addresses, values unused: their W row is zero and y is zeroed).
    auto load_fwd = [&](bool lastb, RpStep<T>& d) {
        const u32 su = t0 ? zsu : (lastb ? 0u : zsL);
        d.zf[0] = *reinterpret_cast<const double*>(zB + zo0);
        d.zf[1] = *reinterpret_cast<const double*>(zB + zo1);
        d.zf[2] = *reinterpret_cast<const double*>(zB + (zo0 + su));
        d.zf[3] = *reinterpret_cast<const double*>(zB + (zo1 + su));
        d.ihl0 = hB[0]; d.ihl1 = hB[lastb ? 0 : 1];
#pragma unroll
        for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(wB + wo[c]);
        const bool clamp = (!t0) && lastb;
        d.S = *reinterpret_cast<const T*>(sB + (clamp ? so - ss : so));
#pragma unroll
        for (int t = 0; t < 6; ++t) d.E[t] = *reinterpret_cast<const T*>(eB + (clamp ? eo[t] - es[t] : eo[t]));
        zB += csL * 8; hB += 1; wB += wstep * (i64)sizeof(T); so += ss;
#pragma unroll
        for (int t = 0; t < 6; ++t) eo[t] += es[t];
    };

    T zprev = Zero<T>::v();
    u32 sto = (u32)(ob[0] * (i64)sizeof(T));       // store cursor (the row's own unknown)
    char* const eW = reinterpret_cast<char*>((a.e + boff_));
    auto fwd_step = [&](bool lastb, const RpStep<T>& cur) {
        const bool full = t0 || !lastb;
        const double ihLm = cur.ihl0;
        const double kL0 = 0.5 * ihLm, kL1 = 0.5 * cur.ihl1;
        const double rs0 = cur.zf[0] + cur.zf[1], rs1 = cur.zf[2] + cur.zf[3];
        const double cs0 = cur.zf[0] + cur.zf[2], cs1 = cur.zf[1] + cur.zf[3];
        const double g0 = (t0 ? K[0] : K[0] * kL1) * rs1;
        const double g1 = (t0 ? K[1] : K[1] * kL0) * rs0;
        T y = cur.S;
        y += g0 * cur.E[0];
        y += g1 * cur.E[1];
        y += (K[2] * cs1) * cur.E[2];
        y += (K[3] * cs0) * cur.E[3];
        y += (K[4] * cs1) * cur.E[4];
        y += (K[5] * cs0) * cur.E[5];
        // coupling to the previous block (zprev = 0 at i = 0): row 0 gets
        // sum_k a_k z_k (ca = 0 on row 0 and on the mirror lanes), row k gets
        // d_k z_k.  ONE exchange per block: every lane publishes
        //   b'_r = b_r - d_r z_r   and   u_r = a_r z_r,
        // then z_r = sum_c W[r][c] b'_c - W[r][0] (u_1 + u_2 + u_3 + u_4).
        const double cz = rs0 * ihLm;
        y += ((tmask * kL0) * cz) * zprev;
        if (!full) y = Zero<T>::v();
        xy[lane] = y;
        xu[lane] = (ca * cz) * zprev;
        const T y0 = xy[g], y1 = xy[LPW + g], y2 = xy[2 * LPW + g], y3 = xy[3 * LPW + g],
                y4 = xy[4 * LPW + g];
        const T u1 = xu[LPW + g], u2 = xu[2 * LPW + g], u3 = xu[3 * LPW + g],
                u4 = xu[4 * LPW + g];
        const T su = (u1 + u2) + (u3 + u4);
        const T z = ((cur.W[0] * (y0 - su) + cur.W[1] * y1) + (cur.W[2] * y2 + cur.W[3] * y3)) + cur.W[4] * y4;
        if (full && rowact) *reinterpret_cast<T*>(eW + sto) = z;      // park z_i in the unknown itself
        sto += ss;
        zprev = z;
    };
    {
#if EMG_RP_STAGES == 3
        // Three register buffers, loop unrolled by three: the loads of blocks
        // i+1 and i+2 are in flight while block i computes.
        RpStep<T> bufA, bufB, bufC;
        i64 i = 0;
        load_fwd(nL == 1, bufA);
        if (nL >= 2) {
            load_fwd(nL == 2, bufB);
            for (; i + 4 < nL; i += 3) {
                load_fwd(false, bufC);                 // block i+2
                fwd_step(false, bufA);                 // block i
                load_fwd(false, bufA);                 // block i+3
                fwd_step(false, bufB);                 // block i+1
                load_fwd(i + 4 == nL - 1, bufB);       // block i+4
                fwd_step(false, bufC);                 // block i+2
            }
            fwd_step(false, bufA);
            fwd_step(i + 1 == nL - 1, bufB);
            for (i64 k = i + 2; k < nL; ++k) {
                load_fwd(k == nL - 1, bufC);
                fwd_step(k == nL - 1, bufC);
            }
        } else {
            fwd_step(true, bufA);
        }
#else
        // Ping-pong register buffers, loop unrolled by two: the loads of block
        // i+1 are in flight while block i computes.  The main loop contains
        // unconditional loads only (counted vmcnt everywhere).
        RpStep<T> bufA, bufB;
        load_fwd(nL == 1, bufA);
        i64 i = 0;
        for (; i + 2 < nL; i += 2) {
            load_fwd(false, bufB);
            fwd_step(false, bufA);
            load_fwd(i + 2 == nL - 1, bufA);
            fwd_step(false, bufB);
        }
        if (i + 1 < nL) {            // two blocks left: i (in A) and i+1 = last
            load_fwd(true, bufB);
            fwd_step(false, bufA);
            fwd_step(true, bufB);
        } else {                     // one block left: the last one (in A)
            fwd_step(true, bufA);
        }
#endif
    }

    // ----------------------------- backward --------------------------------
    // x_{nL-1} = z_{nL-1}; zprev holds x_{i+1} (transverse rows: 0 for the last block)
    if (nL >= 2) {
        const char* qW = reinterpret_cast<const char*>(a.fac) + (nL - 2) * wstep * (i64)sizeof(T);
        const char* qz = reinterpret_cast<const char*>(a.zeta) + (nL - 1) * csL * 8;
        const double* qH = a.ih[L] + (nL - 1);
        u32 qo = (u32)((ob[0] + (nL - 2) * os[0]) * (i64)sizeof(T));   // z_i / x_i of this row
        auto load_bwd = [&](RpBack<T>& d) {
#pragma unroll
            for (int c = 0; c < 5; ++c) d.W[c] = *reinterpret_cast<const T*>(qW + wo[c]);
            d.zi = *reinterpret_cast<const T*>(eB + qo);
            d.p0 = *reinterpret_cast<const double*>(qz + zo0);
            d.p1 = *reinterpret_cast<const double*>(qz + zo1);
            d.ihln = *qH;
            qW -= wstep * (i64)sizeof(T); qo -= ss; qz -= csL * 8; qH -= 1;
        };
        u32 qs = (u32)((ob[0] + (nL - 2) * os[0]) * (i64)sizeof(T));
        auto bwd_step = [&](bool nextlast, const RpBack<T>& bc) {
            const double ihLn = bc.ihln;
            const double cz = (bc.p0 + bc.p1) * ihLn;
            const double dm = nextlast ? 0.0 : tmask;    // next block is the last: no d-coupling
            // ONE exchange: lane c publishes Q_c = d_c x_c (lane 0: x_0) and its
            // real a_c; then v_c = a_c x_0 + Q_c, w_r = sum_{c>=1} W[r][c] v_c.
            const double ac = ca * cz;
            const double dc = ((-0.5 * dm) * ihLn) * cz;
            xy[lane] = t0 ? zprev : dc * zprev;
            T aa = Zero<T>::v();
            add_real(aa, ac);
            xu[lane] = aa;
            const T x0 = xy[g];
            const T v1 = real_of(xu[LPW + g]) * x0 + xy[LPW + g];
            const T v2 = real_of(xu[2 * LPW + g]) * x0 + xy[2 * LPW + g];
            const T v3 = real_of(xu[3 * LPW + g]) * x0 + xy[3 * LPW + g];
            const T v4 = real_of(xu[4 * LPW + g]) * x0 + xy[4 * LPW + g];
            const T w = (bc.W[1] * v1 + bc.W[2] * v2) + (bc.W[3] * v3 + bc.W[4] * v4);
            const T x = bc.zi - w;
            if (rowact) *reinterpret_cast<T*>(eW + qs) = x;
            qs -= ss;
            zprev = x;
        };
#if EMG_RP_STAGES == 3
        RpBack<T> bA, bB, bC;
        i64 i = nL - 2;
        load_bwd(bA);                                  // block i
        if (i >= 1) {
            load_bwd(bB);                              // block i-1
            bool first = true;
            for (; i >= 4; i -= 3) {
                load_bwd(bC);                          // block i-2
                bwd_step(first, bA);                   // block i
                load_bwd(bA);                          // block i-3
                bwd_step(false, bB);                   // block i-1
                load_bwd(bB);                          // block i-4
                bwd_step(false, bC);                   // block i-2
                first = false;
            }
            bwd_step(first, bA);
            bwd_step(false, bB);
            for (i64 k = i - 2; k >= 0; --k) {
                load_bwd(bC);
                bwd_step(false, bC);
            }
        } else {
            bwd_step(true, bA);
        }
#else
        RpBack<T> bA, bB;
        load_bwd(bA);
        i64 i = nL - 2;
        if (i >= 2) {                // peeled first pair (the only one with nextlast)
            load_bwd(bB);
            bwd_step(true, bA);
            load_bwd(bA);
            bwd_step(false, bB);
            i -= 2;
            for (; i >= 2; i -= 2) {
                load_bwd(bB);
                bwd_step(false, bA);
                load_bwd(bA);
                bwd_step(false, bB);
            }
            if (i == 1) {
                load_bwd(bB);
                bwd_step(false, bA);
                bwd_step(false, bB);
            } else {
                bwd_step(false, bA);
            }
        } else if (i == 1) {
            load_bwd(bB);
            bwd_step(true, bA);
            bwd_step(false, bB);
        } else {
            bwd_step(true, bA);
        }
#endif
    }
}

template <class T>
struct PointArgs {
    i64 nC[3];
    FieldLayout fl;
    T* e;
    const T* s;
    const T* eta[3];
    const double* zeta;
    const double* h[3];
    const double* ih[3];   // 1 / h, correctly rounded (qdiv)
    Batch bt;
    int mode;
    int col;           // mode 0: colour bits (x | y<<1 | z<<2)
    i64 cnt[3];        // mode 0: nodes per axis in this colour
    i64 t;             // mode 1
};

// x / h from the cached reciprocal ih = RN(1 / h): q = RN(x ih), r = x - q h (exact, FMA), RN(q + r ih) is the correctly
// rounded quotient (Markstein's theorem; h is a cell width: no special values).
__device__ __forceinline__ double qdiv(double x, double h, double ih) {
    const double q = x * ih;
    const double r = fma(-q, h, x);
    return fma(r, ih, q);
}
__device__ __forceinline__ c128 qdiv(c128 x, double h, double ih) { return mk(qdiv(x.re, h, ih), qdiv(x.im, h, ih)); }

template <class T>
__global__ __launch_bounds__(EMG_LINE_BLOCK) void k_point_sweep(PointArgs<T> a) {
    EMG_ARGS_BURST(EMG_S(a.nC[0]), EMG_S(a.nC[1]), EMG_S(a.nC[2]), EMG_S(a.e), EMG_S(a.s), EMG_S(a.eta[0]), EMG_S(a.eta[1]), EMG_S(a.eta[2]),
                   EMG_S(a.zeta), EMG_S(a.h[0]), EMG_S(a.h[1]), EMG_S(a.h[2]), EMG_S(a.ih[0]), EMG_S(a.ih[1]), EMG_S(a.ih[2]), EMG_S(a.bt.st),
                   EMG_S(a.bt.mask), EMG_S(a.mode), EMG_S(a.col), EMG_S(a.cnt[0]), EMG_S(a.cnt[1]), EMG_S(a.cnt[2]), EMG_S(a.t),
                   EMG_S(a.fl.off[1]), EMG_S(a.fl.off[2]));
    EMG_BATCH(y, a.bt);
    const i64 nx = a.nC[0], ny = a.nC[1], nz = a.nC[2];
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 ix, iy, iz;
    if (a.mode == 0) {
        if (idx >= a.cnt[0] * a.cnt[1] * a.cnt[2]) return;
        i64 qx, qy, qz;
        unlin3(idx, a.cnt[0], a.cnt[1], a.cnt[2], qx, qy, qz);
        ix = 1 + (a.col & 1) + 2 * qx;
        iy = 1 + ((a.col >> 1) & 1) + 2 * qy;
        iz = 1 + ((a.col >> 2) & 1) + 2 * qz;
    } else {
        if (idx >= (ny - 1) * (nz - 1)) return;
        unlin2(idx, ny - 1, iy, iz);
        iy += 1; iz += 1;
        ix = a.t - 2 * iy - 4 * iz;
        if (ix < 1 || ix > nx - 1) return;
    }
    const FieldLayout& f = a.fl;
    T* e = (a.e + boff_);
    const T* s = (a.s + boff_);
#define PX(i, j, k) (f.off[0] + (i) * f.st[0][0] + (j) * f.st[0][1] + (k) * f.st[0][2])
#define PY(i, j, k) (f.off[1] + (i) * f.st[1][0] + (j) * f.st[1][1] + (k) * f.st[1][2])
#define PZ(i, j, k) (f.off[2] + (i) * f.st[2][0] + (j) * f.st[2][1] + (k) * f.st[2][2])
#define CI(i, j, k) ((i) + nx * ((j) + ny * (k)))
    const i64 ixm = ix - 1, ixp = ix + 1, iym = iy - 1, iyp = iy + 1, izm = iz - 1, izp = iz + 1;
    const double hx[2] = {a.h[0][ixm], a.h[0][ix]}, hy[2] = {a.h[1][iym], a.h[1][iy]},
                 hz[2] = {a.h[2][izm], a.h[2][iz]};
    // The 84 divisions by a cell width per node (core.py:322-463) are evaluated as qdiv(x, h, 1/h): one product and two
    // FMAs with the cached, correctly rounded reciprocal give the correctly rounded quotient (Markstein) -- the bits of
    // x / h at an eighth of the instructions (a double-precision division expands to ~25).  0.5 / h = 0.5 * (1/h) exactly.
    const double ihx[2] = {a.ih[0][ixm], a.ih[0][ix]}, ihy[2] = {a.ih[1][iym], a.ih[1][iy]},
                 ihz[2] = {a.ih[2][izm], a.ih[2][iz]};
    const double kx[2] = {0.5 * ihx[0], 0.5 * ihx[1]}, ky[2] = {0.5 * ihy[0], 0.5 * ihy[1]},
                 kz[2] = {0.5 * ihz[0], 0.5 * ihz[1]};
    double z[2][2][2];
    T etx[2][2][2], ety[2][2][2], etz[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const i64 p = CI(ixm + i, iym + j, izm + k);
                z[i][j][k] = a.zeta[p];
                etx[i][j][k] = a.eta[0][p];
                ety[i][j][k] = a.eta[1][p];
                etz[i][j][k] = a.eta[2][p];
            }
    // the 24 neighbour values the six right-hand sides use (each twice), loaded TOGETHER before the arithmetic: written
    // where they are used, the compiler issued one load per use and waited for each (45 dependent round trips)
#define IXI_ixm 0
#define IXI_ix 1
#define IXI_ixp 2
#define IYI_iym 0
#define IYI_iy 1
#define IYI_iyp 2
#define IZI_izm 0
#define IZI_iz 1
#define IZI_izp 2
    T exl[3][3][3], eyl[3][3][3], ezl[3][3][3];
    exl[IXI_ix][IYI_iy][IZI_izm] = e[PX(ix, iy, izm)];
    exl[IXI_ix][IYI_iy][IZI_izp] = e[PX(ix, iy, izp)];
    exl[IXI_ix][IYI_iym][IZI_iz] = e[PX(ix, iym, iz)];
    exl[IXI_ix][IYI_iyp][IZI_iz] = e[PX(ix, iyp, iz)];
    exl[IXI_ixm][IYI_iy][IZI_izm] = e[PX(ixm, iy, izm)];
    exl[IXI_ixm][IYI_iy][IZI_izp] = e[PX(ixm, iy, izp)];
    exl[IXI_ixm][IYI_iym][IZI_iz] = e[PX(ixm, iym, iz)];
    exl[IXI_ixm][IYI_iyp][IZI_iz] = e[PX(ixm, iyp, iz)];
    eyl[IXI_ix][IYI_iy][IZI_izm] = e[PY(ix, iy, izm)];
    eyl[IXI_ix][IYI_iy][IZI_izp] = e[PY(ix, iy, izp)];
    eyl[IXI_ix][IYI_iym][IZI_izm] = e[PY(ix, iym, izm)];
    eyl[IXI_ix][IYI_iym][IZI_izp] = e[PY(ix, iym, izp)];
    eyl[IXI_ixm][IYI_iy][IZI_iz] = e[PY(ixm, iy, iz)];
    eyl[IXI_ixm][IYI_iym][IZI_iz] = e[PY(ixm, iym, iz)];
    eyl[IXI_ixp][IYI_iy][IZI_iz] = e[PY(ixp, iy, iz)];
    eyl[IXI_ixp][IYI_iym][IZI_iz] = e[PY(ixp, iym, iz)];
    ezl[IXI_ix][IYI_iym][IZI_iz] = e[PZ(ix, iym, iz)];
    ezl[IXI_ix][IYI_iym][IZI_izm] = e[PZ(ix, iym, izm)];
    ezl[IXI_ix][IYI_iyp][IZI_iz] = e[PZ(ix, iyp, iz)];
    ezl[IXI_ix][IYI_iyp][IZI_izm] = e[PZ(ix, iyp, izm)];
    ezl[IXI_ixm][IYI_iy][IZI_iz] = e[PZ(ixm, iy, iz)];
    ezl[IXI_ixm][IYI_iy][IZI_izm] = e[PZ(ixm, iy, izm)];
    ezl[IXI_ixp][IYI_iy][IZI_iz] = e[PZ(ixp, iy, iz)];
    ezl[IXI_ixp][IYI_iy][IZI_izm] = e[PZ(ixp, iy, izm)];
    __builtin_amdgcn_sched_barrier(0);
    // core.py:322-345.  z[x][y][z], 0 = minus, 1 = plus.
    const double mzyLxm = ky[0] * (z[0][0][1] + z[0][0][0]);
    const double mzyRxm = ky[1] * (z[0][1][1] + z[0][1][0]);
    const double myzLxm = kz[0] * (z[0][1][0] + z[0][0][0]);
    const double myzRxm = kz[1] * (z[0][1][1] + z[0][0][1]);
    const double mzyLxp = ky[0] * (z[1][0][1] + z[1][0][0]);
    const double mzyRxp = ky[1] * (z[1][1][1] + z[1][1][0]);
    const double myzLxp = kz[0] * (z[1][1][0] + z[1][0][0]);
    const double myzRxp = kz[1] * (z[1][1][1] + z[1][0][1]);
    const double mzxLym = kx[0] * (z[0][0][1] + z[0][0][0]);
    const double mzxRym = kx[1] * (z[1][0][1] + z[1][0][0]);
    const double mxzLym = kz[0] * (z[1][0][0] + z[0][0][0]);
    const double mxzRym = kz[1] * (z[1][0][1] + z[0][0][1]);
    const double mzxLyp = kx[0] * (z[0][1][1] + z[0][1][0]);
    const double mzxRyp = kx[1] * (z[1][1][1] + z[1][1][0]);
    const double mxzLyp = kz[0] * (z[1][1][0] + z[0][1][0]);
    const double mxzRyp = kz[1] * (z[1][1][1] + z[0][1][1]);
    const double myxLzm = kx[0] * (z[0][1][0] + z[0][0][0]);
    const double myxRzm = kx[1] * (z[1][1][0] + z[1][0][0]);
    const double mxyLzm = ky[0] * (z[1][0][0] + z[0][0][0]);
    const double mxyRzm = ky[1] * (z[1][1][0] + z[0][1][0]);
    const double myxLzp = kx[0] * (z[0][1][1] + z[0][0][1]);
    const double myxRzp = kx[1] * (z[1][1][1] + z[1][0][1]);
    const double mxyLzp = ky[0] * (z[1][0][1] + z[0][0][1]);
    const double mxyRzp = ky[1] * (z[1][1][1] + z[0][1][1]);

    // A[r][c] lower triangle (core.py:364-401)
    T A[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] = Zero<T>::v();
    A[0][0] = -((((etx[0][1][1] + etx[0][1][0]) + etx[0][0][1]) + etx[0][0][0]) * 0.25);
    A[1][1] = -((((etx[1][1][1] + etx[1][1][0]) + etx[1][0][1]) + etx[1][0][0]) * 0.25);
    A[2][2] = -((((ety[1][0][1] + ety[1][0][0]) + ety[0][0][1]) + ety[0][0][0]) * 0.25);
    A[3][3] = -((((ety[1][1][1] + ety[1][1][0]) + ety[0][1][1]) + ety[0][1][0]) * 0.25);
    A[4][4] = -((((etz[1][1][0] + etz[1][0][0]) + etz[0][1][0]) + etz[0][0][0]) * 0.25);
    A[5][5] = -((((etz[1][1][1] + etz[1][0][1]) + etz[0][1][1]) + etz[0][0][1]) * 0.25);
    add_real(A[0][0], qdiv(mzyRxm, hy[1], ihy[1]) + qdiv(mzyLxm, hy[0], ihy[0]));
    add_real(A[0][0], qdiv(myzRxm, hz[1], ihz[1]) + qdiv(myzLxm, hz[0], ihz[0]));
    add_real(A[1][1], qdiv(mzyRxp, hy[1], ihy[1]) + qdiv(mzyLxp, hy[0], ihy[0]));
    add_real(A[1][1], qdiv(myzRxp, hz[1], ihz[1]) + qdiv(myzLxp, hz[0], ihz[0]));
    add_real(A[2][2], qdiv(mzxRym, hx[1], ihx[1]) + qdiv(mzxLym, hx[0], ihx[0]));
    add_real(A[2][2], qdiv(mxzRym, hz[1], ihz[1]) + qdiv(mxzLym, hz[0], ihz[0]));
    add_real(A[3][3], qdiv(mzxRyp, hx[1], ihx[1]) + qdiv(mzxLyp, hx[0], ihx[0]));
    add_real(A[3][3], qdiv(mxzRyp, hz[1], ihz[1]) + qdiv(mxzLyp, hz[0], ihz[0]));
    add_real(A[4][4], qdiv(myxRzm, hx[1], ihx[1]) + qdiv(myxLzm, hx[0], ihx[0]));
    add_real(A[4][4], qdiv(mxyRzm, hy[1], ihy[1]) + qdiv(mxyLzm, hy[0], ihy[0]));
    add_real(A[5][5], qdiv(myxRzp, hx[1], ihx[1]) + qdiv(myxLzp, hx[0], ihx[0]));
    add_real(A[5][5], qdiv(mxyRzp, hy[1], ihy[1]) + qdiv(mxyLzp, hy[0], ihy[0]));
    add_real(A[2][0], qdiv(-mzyLxm, hx[0], ihx[0]));
    add_real(A[3][0], qdiv(mzyRxm, hx[0], ihx[0]));
    add_real(A[4][0], qdiv(-myzLxm, hx[0], ihx[0]));
    add_real(A[5][0], qdiv(myzRxm, hx[0], ihx[0]));
    add_real(A[2][1], qdiv(mzyLxp, hx[1], ihx[1]));
    add_real(A[3][1], qdiv(-mzyRxp, hx[1], ihx[1]));
    add_real(A[4][1], qdiv(myzLxp, hx[1], ihx[1]));
    add_real(A[5][1], qdiv(-myzRxp, hx[1], ihx[1]));
    add_real(A[4][2], qdiv(-mxzLym, hy[0], ihy[0]));
    add_real(A[5][2], qdiv(mxzRym, hy[0], ihy[0]));
    add_real(A[4][3], qdiv(mxzLyp, hy[1], ihy[1]));
    add_real(A[5][3], qdiv(-mxzRyp, hy[1], ihy[1]));

    // rhs (core.py:407-463)
    T b[6];
    b[0] = s[PX(ixm, iy, iz)]; b[1] = s[PX(ix, iy, iz)];
    b[2] = s[PY(ix, iym, iz)]; b[3] = s[PY(ix, iy, iz)];
    b[4] = s[PZ(ix, iy, izm)]; b[5] = s[PZ(ix, iy, iz)];
#define EX(i, j, k) exl[IXI_##i][IYI_##j][IZI_##k]
#define EY(i, j, k) eyl[IXI_##i][IYI_##j][IZI_##k]
#define EZ(i, j, k) ezl[IXI_##i][IYI_##j][IZI_##k]
    b[0] += mzyRxm * (qdiv(EY(ixm, iy, iz), hx[0], ihx[0]) + qdiv(EX(ixm, iyp, iz), hy[1], ihy[1]));
    b[0] += mzyLxm * (qdiv(-EY(ixm, iym, iz), hx[0], ihx[0]) + qdiv(EX(ixm, iym, iz), hy[0], ihy[0]));
    b[0] += myzRxm * (qdiv(EZ(ixm, iy, iz), hx[0], ihx[0]) + qdiv(EX(ixm, iy, izp), hz[1], ihz[1]));
    b[0] += myzLxm * (qdiv(-EZ(ixm, iy, izm), hx[0], ihx[0]) + qdiv(EX(ixm, iy, izm), hz[0], ihz[0]));

    b[1] += mzyRxp * (qdiv(-EY(ixp, iy, iz), hx[1], ihx[1]) + qdiv(EX(ix, iyp, iz), hy[1], ihy[1]));
    b[1] += mzyLxp * (qdiv(EY(ixp, iym, iz), hx[1], ihx[1]) + qdiv(EX(ix, iym, iz), hy[0], ihy[0]));
    b[1] += myzRxp * (qdiv(-EZ(ixp, iy, iz), hx[1], ihx[1]) + qdiv(EX(ix, iy, izp), hz[1], ihz[1]));
    b[1] += myzLxp * (qdiv(EZ(ixp, iy, izm), hx[1], ihx[1]) + qdiv(EX(ix, iy, izm), hz[0], ihz[0]));

    b[2] += mzxRym * (qdiv(EY(ixp, iym, iz), hx[1], ihx[1]) + qdiv(EX(ix, iym, iz), hy[0], ihy[0]));
    b[2] += mzxLym * (qdiv(EY(ixm, iym, iz), hx[0], ihx[0]) - qdiv(EX(ixm, iym, iz), hy[0], ihy[0]));
    b[2] += mxzRym * (qdiv(EZ(ix, iym, iz), hy[0], ihy[0]) + qdiv(EY(ix, iym, izp), hz[1], ihz[1]));
    b[2] += mxzLym * (qdiv(-EZ(ix, iym, izm), hy[0], ihy[0]) + qdiv(EY(ix, iym, izm), hz[0], ihz[0]));

    b[3] += mzxRyp * (qdiv(EY(ixp, iy, iz), hx[1], ihx[1]) - qdiv(EX(ix, iyp, iz), hy[1], ihy[1]));
    b[3] += mzxLyp * (qdiv(EY(ixm, iy, iz), hx[0], ihx[0]) + qdiv(EX(ixm, iyp, iz), hy[1], ihy[1]));
    b[3] += mxzRyp * (qdiv(-EZ(ix, iyp, iz), hy[1], ihy[1]) + qdiv(EY(ix, iy, izp), hz[1], ihz[1]));
    b[3] += mxzLyp * (qdiv(EZ(ix, iyp, izm), hy[1], ihy[1]) + qdiv(EY(ix, iy, izm), hz[0], ihz[0]));

    b[4] += myxRzm * (qdiv(EZ(ixp, iy, izm), hx[1], ihx[1]) + qdiv(EX(ix, iy, izm), hz[0], ihz[0]));
    b[4] += myxLzm * (qdiv(EZ(ixm, iy, izm), hx[0], ihx[0]) - qdiv(EX(ixm, iy, izm), hz[0], ihz[0]));
    b[4] += mxyRzm * (qdiv(EZ(ix, iyp, izm), hy[1], ihy[1]) + qdiv(EY(ix, iy, izm), hz[0], ihz[0]));
    b[4] += mxyLzm * (qdiv(EZ(ix, iym, izm), hy[0], ihy[0]) - qdiv(EY(ix, iym, izm), hz[0], ihz[0]));

    b[5] += myxRzp * (qdiv(EZ(ixp, iy, iz), hx[1], ihx[1]) - qdiv(EX(ix, iy, izp), hz[1], ihz[1]));
    b[5] += myxLzp * (qdiv(EZ(ixm, iy, iz), hx[0], ihx[0]) + qdiv(EX(ixm, iy, izp), hz[1], ihz[1]));
    b[5] += mxyRzp * (qdiv(EZ(ix, iyp, iz), hy[1], ihy[1]) - qdiv(EY(ix, iy, izp), hz[1], ihz[1]));
    b[5] += mxyLzp * (qdiv(EZ(ix, iym, iz), hy[0], ihy[0]) + qdiv(EY(ix, iym, izp), hz[1], ihz[1]));

    // dense 6x6 LDL^T without pivoting (= core.solve for n = 6, core.py:466)
    T D[6], Dinv[6], Lm[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        T dj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= (Lm[j][k] * Lm[j][k]) * D[k];
        D[j] = dj;
        Dinv[j] = recip(dj);
#pragma unroll
        for (int r = j + 1; r < 6; ++r) {
            T v = A[r][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= (Lm[r][k] * Lm[j][k]) * D[k];
            Lm[r][j] = v * Dinv[j];
        }
    }
#pragma unroll
    for (int j = 1; j < 6; ++j) {
        T hsum = Zero<T>::v();
#pragma unroll
        for (int k = 0; k < j; ++k) hsum += Lm[j][k] * b[k];
        b[j] -= hsum;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) b[j] = b[j] * Dinv[j];
#pragma unroll
    for (int j = 4; j >= 0; --j) {
        T hsum = Zero<T>::v();
#pragma unroll
        for (int k = j + 1; k < 6; ++k) hsum += Lm[k][j] * b[k];
        b[j] -= hsum;
    }
    e[PX(ixm, iy, iz)] = b[0]; e[PX(ix, iy, iz)] = b[1];
    e[PY(ix, iym, iz)] = b[2]; e[PY(ix, iy, iz)] = b[3];
    e[PZ(ix, iy, izm)] = b[4]; e[PZ(ix, iy, iz)] = b[5];
#undef EX
#undef EY
#undef EZ
#undef PX
#undef PY
#undef PZ
#undef CI
}

// ---------------------------------------------------------------------------
// core.solve / core.blocks_to_amat as single-thread device kernels: the
// reference exposes them, its smoothers are their only callers; kept for the
// drop-in `core` surface and the known-answer tests.
// ---------------------------------------------------------------------------
template <class T>
__global__ void k_solve_banded(T* amat, T* bvec, i64 n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    T hh;
    T d = recip(amat[0]);
    for (i64 i = 1; i < (n < 6 ? n : 6); ++i) amat[i] = amat[i] * d;
    for (i64 j = 1; j < n; ++j) {
        hh = Zero<T>::v();
        for (i64 k = (j - 5 > 0 ? j - 5 : 0); k < j; ++k) hh += (amat[j + 5 * k] * amat[j + 5 * k]) * amat[6 * k];
        amat[6 * j] -= hh;
        d = recip(amat[6 * j]);
        for (i64 i = j + 1; i < (n < j + 6 ? n : j + 6); ++i) {
            hh = Zero<T>::v();
            for (i64 k = (i - 5 > 0 ? i - 5 : 0); k < j; ++k) hh += (amat[i + 5 * k] * amat[j + 5 * k]) * amat[6 * k];
            amat[i + 5 * j] -= hh;
            amat[i + 5 * j] = amat[i + 5 * j] * d;
        }
    }
    amat[6 * (n - 1)] = d;
    for (i64 j = n - 2; j >= 0; --j) amat[6 * j] = recip(amat[6 * j]);
    for (i64 j = 1; j < n; ++j) {
        hh = Zero<T>::v();
        for (i64 k = (j - 5 > 0 ? j - 5 : 0); k < j; ++k) hh += amat[j + 5 * k] * bvec[k];
        bvec[j] -= hh;
    }
    for (i64 j = 0; j < n; ++j) bvec[j] = bvec[j] * amat[6 * j];
    for (i64 j = n - 2; j >= 0; --j) {
        hh = Zero<T>::v();
        for (i64 k = j + 1; k < (n < j + 6 ? n : j + 6); ++k) hh += amat[k + 5 * j] * bvec[k];
        bvec[j] -= hh;
    }
}

template <class T>
__global__ void k_blocks_to_amat(T* amat, T* bvec, const T* middle, const double* left, const T* rhs,
                                 i64 im, i64 nC) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const i64 fam = 5 * im, mam = fam - 5;
    if (im == 0) {
        for (int k = 0; k < 5; ++k) bvec[k] = rhs[k];
        for (int k = 0; k < 5; ++k)
            for (int m = 0; m <= k; ++m) amat[k + 5 * m] = middle[k + 5 * m];
    } else if (im <= nC - 2 && nC > 2) {
        for (int k = 0; k < 5; ++k) bvec[k + fam] = rhs[k];
        for (int m = 1; m < 5; ++m)
            for (int k = 0; k <= m; ++k) {
                T v = Zero<T>::v();
                add_real(v, left[k + 5 * m]);
                amat[k + fam + 5 * (m + mam)] = v;
            }
        for (int k = 0; k < 5; ++k)
            for (int m = 0; m <= k; ++m) amat[k + fam + 5 * (m + fam)] = middle[k + 5 * m];
    } else if (im == nC - 1) {
        bvec[fam] = rhs[0];
        for (int m = 1; m < 5; ++m) {
            T v = Zero<T>::v();
            add_real(v, left[5 * m]);
            amat[fam + 5 * (m + mam)] = v;
        }
        amat[6 * fam] = middle[0];
    }
}
