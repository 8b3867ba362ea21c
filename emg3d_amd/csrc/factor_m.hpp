// The MIRRORED two-sided line factorisation (k_line_factor_m) that k_line_sweep_thm sweeps on: the same line solves as
// the one-sided factorisation of smooth.hpp (reference emg3d/core.py:477-1316, core.py:1447-1582), half the chain length.
//
// A line holds the unknowns l_0 .. l_{n-1} (edges along the line) and T_0 .. T_{n-2} (the four transverse edges
// at node i+1).  The reference eliminates them in the natural order l_0, T_0, l_1, T_1, ...  A two-sided
// elimination that groups the unknowns of the right half like the left half -- [l_i; T_i], processed downwards --
// loses 3-4 digits on ill-conditioned lines (lines inside a resistive body: every interior node carries a
// discrete gradient that only eta regularises; measured 1e-8 instead of 2e-12 against 80-bit arithmetic,
// tests/tools/conditioning.py; round 1's k_line_sweep_th).  The MIRROR image of the natural order does not:
//     left  blocks [l_i; T_i],     i = 0 .. m-1,      eliminated upwards   (as the reference does),
//     right blocks [l_j; T_{j-1}], j = n-1 .. m+2,    eliminated downwards (the reference's order on the reversed line),
//     middle       [l_m; T_m; l_{m+1}]                (6 unknowns) last,
// is as accurate as the one-sided order (2e-12 on the same lines).  In the mirrored grouping the right half runs the
// SAME recurrences as the left half on a reversed index with the sign of the l-T coupling flipped (u -> -u).
//
// Factor layout [block slot][entry 0..14][line]: slot i < m: W of the left block i; slot j > m+1: W of the right
// block j = [l_j; T_{j-1}]; slots m and m+1: the 21 entries of the symmetric 6x6 middle inverse (unknown order
// l_m, T_m[0..3], l_{m+1}; packed lower triangle p = r (r + 1) / 2 + c; p < 15 in slot m, p - 15 in slot m+1).
#pragma once
#include "smooth.hpp"

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
