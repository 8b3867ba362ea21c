// Producer / chain line sweep: the dependent recurrence stripped to 16 multiply-adds per block
// (reference emg3d/core.py:477-1316 line solves; same cached block factorisation and the same affine-map form of the
// two recurrences as smooth_qpl.hpp).
//
//   forward : u_i = c_i + G_i u_{i-1},  u = z[1..4],   c_i = (W_i b_i)[1..4],  G_i = -(W_i A_i)[1..4][1..4]
//   backward: v_i = g_i + H_i v_{i+1},  v = A_i^T x_i, g_i = A_i^T z_i,        H_i = -(A_i^T W_i)[1..4][1..4]
//
// Everything but these two 4-vector recurrences is independent from block to block: the right-hand side b_i (30 neighbour
// values, the zeta face), c_i, G_i, and for the way back g_i, H_i.  Even z_i[0] = (W_i b_i)[0] + G0_i u_{i-1} and
// x_i = z_i - W_i[.][1..4] v_{i+1} are "one more row of the same step": an affine function of the chain vector that nobody
// feeds back.  The chain kernels (k_line_sweep_rp / _thm / _qc) evaluate all of it inside the dependent step (270-380
// instructions per block on the one wave that carries the line: that wave's instruction stream IS the launch time); the scan
// kernel (k_line_sweep_qpl) makes the chain parallel at 4 x the arithmetic.  Here a workgroup of four waves carries NL <= 4
// lines and splits the work by ROLE, a workgroup barrier per chunk ("tick") of 36 / NL blocks of each line:
//
//   waves 1..3     producers, lane = (block, row): five lanes per 5x5 block (k_line_sweep_rp's per-row right-hand side), 12
//                  blocks per wave and tick: loads, right-hand side, c_i and the rows of G_i -> LDS "stage" (double
//                  buffered).  Three register sets in rotation: the loads of the next two ticks are in flight;
//   wave 0         the chain, lane = (line, row): the rows of a line in the lanes of one 16-lane DPP row.  A step reads its
//                  row of (c_i, G_i) from the stage the producers filled in the tick before and is 16
//                  `v_fmac_f64_dpp row_newbcast:k` -- the 64-bit DPP form of gfx90a+ feeds lane k of each row into the
//                  multiply-add itself: no cross-lane moves, no LDS on the dependent path, ~33 instructions per block
//                  whatever the number of lines.  Lanes 0..3 carry u (v on the way back); lane 4 evaluates z_i[0] with the
//                  same 16 instructions, and on the way back lanes 4..8 evaluate x_i[0..4], which the producers store.
//                  u_i and z_i[0] are parked in LDS (never in HBM: the 160 B per block the other chain kernels park in
//                  `e` are gone).
//
// Factor layout: k_line_sweep_qpl's [line][entry][block slots] (consecutive blocks of a line contiguous).
//
// Served: colour-ordered launches on levels without parity-split copies, lines of pc_min_nl .. pc_max_nl blocks
// (MG::pc_lines in mg.hpp); everything else keeps its kernel.
#pragma once
#include "smooth_qpl.hpp"


// acc <- acc + G . u_row: lanes 0..3 of each 16-lane row hold u_0..3; row_newbcast:k feeds lane k's value to every lane of the
// row.  The s_nop covers the VALU-write -> DPP-read hazard (2 wait states) that the compiler cannot see inside an asm block;
// the two accumulators alternate, so a dependent pair is two instructions apart.
#define PC_DPP_ " row_mask:0xf bank_mask:0xf\n"
__device__ __forceinline__ void pc_chain_step(c128& acc, const c128 u, const c128 (&G)[4]) {
    asm("s_nop 1\n"
        "v_fmac_f64_dpp %0, %2, %4 row_newbcast:0" PC_DPP_
        "v_fmac_f64_dpp %1, %3, %4 row_newbcast:0" PC_DPP_
        "v_fmac_f64_dpp %0, %3, -%5 row_newbcast:0" PC_DPP_
        "v_fmac_f64_dpp %1, %2, %5 row_newbcast:0" PC_DPP_
        "v_fmac_f64_dpp %0, %2, %6 row_newbcast:1" PC_DPP_
        "v_fmac_f64_dpp %1, %3, %6 row_newbcast:1" PC_DPP_
        "v_fmac_f64_dpp %0, %3, -%7 row_newbcast:1" PC_DPP_
        "v_fmac_f64_dpp %1, %2, %7 row_newbcast:1" PC_DPP_
        "v_fmac_f64_dpp %0, %2, %8 row_newbcast:2" PC_DPP_
        "v_fmac_f64_dpp %1, %3, %8 row_newbcast:2" PC_DPP_
        "v_fmac_f64_dpp %0, %3, -%9 row_newbcast:2" PC_DPP_
        "v_fmac_f64_dpp %1, %2, %9 row_newbcast:2" PC_DPP_
        "v_fmac_f64_dpp %0, %2, %10 row_newbcast:3" PC_DPP_
        "v_fmac_f64_dpp %1, %3, %10 row_newbcast:3" PC_DPP_
        "v_fmac_f64_dpp %0, %3, -%11 row_newbcast:3" PC_DPP_
        "v_fmac_f64_dpp %1, %2, %11 row_newbcast:3" PC_DPP_
        : "+v"(acc.re), "+v"(acc.im)
        : "v"(u.re), "v"(u.im), "v"(G[0].re), "v"(G[0].im), "v"(G[1].re), "v"(G[1].im), "v"(G[2].re), "v"(G[2].im),
          "v"(G[3].re), "v"(G[3].im));
}
__device__ __forceinline__ void pc_chain_step(double& acc, const double u, const double (&G)[4]) {
    asm("s_nop 1\n"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0" PC_DPP_
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1" PC_DPP_
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2" PC_DPP_
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3" PC_DPP_
        : "+v"(acc)
        : "v"(u), "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]));
}
#undef PC_DPP_

// LDS elements (of T) per workgroup: stage [2 buffers][36 blocks][9 chain rows][5] (forward: 5 rows), exchange
// [3 waves][12 blocks][5], park [nLp blocks][NL][5] (u[0..3], z[0]), 64 where the shadow lanes of the chain put their copies
template <int NL>
__host__ __device__ inline int pc_wg_elems(int nL) {
    const int C = 36 / NL;
    const int nLp = ((nL + C - 1) / C) * C;
    return 2 * 36 * 45 + 180 + nLp * NL * 5 + 64;
}

template <class T>
struct PcFwd {          // what a producer lane loads for its row of one block, forward pass
    T W[5], E[6], S;
    double f[4], n0, n1, ihl0, ihl1;
};
template <class T>
struct PcBwd {          // ... backward pass
    T Wt[4], W0t[4];
    double f[4], ihl0;
};

template <class T, int NL>
__global__ __launch_bounds__(256) void k_line_sweep_pc(LineArgs<T> a) {
    constexpr int C = 36 / NL;                   // blocks (of a line) per tick
    typedef unsigned int u32;
    extern __shared__ __align__(16) unsigned char pc_smem_[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    EMG_SWEEP_WG(a)
    const int nL = (int)a.rs.nL;
    const int dbg = a.tile;       // lab (EMG3D_Q_TILE), timing only: 1 = no chain steps, 4 = no stores, 8 = every lane loads lane 0's addresses
    const int nch = (nL + C - 1) / C;
    const int nLp = nch * C;
    T* const stage = reinterpret_cast<T*>(pc_smem_);
    T* const xb = stage + 2 * 36 * 45;
    T* const park = stage + 2 * 36 * 45 + 180;
    T* const dumpb = park + nLp * NL * 5;
    const u32 nlines = (u32)(a.cntA * a.cntB);
    const u32 wline0 = (u32)wg * NL;
    if (wline0 >= nlines) return;                // (whole workgroup)

    // ---- producer lane (waves 1..3; wave 0 computes the same values for its lanes and ignores them): row r (0: the edge
    //      along the line; 1,2 / 3,4: the P- / Q-directed edges at node i+1) of block bsub (of the tick) of line g; lanes
    //      60..63 mirror lane 0 (no LDS writes, no stores) ----
    const int pw = wv > 0 ? wv - 1 : 0;
    const bool pact = lane < 60;
    const int pl = pact ? lane : 0;
    const int r = pl / 12, it = pw * 12 + (pl - 12 * r);        // it: the block's slot in the stage, 0..35
    const int g = it % NL, bsub = it / NL;
    const bool live = wline0 + (u32)g < nlines;
    const u32 gidx = live ? wline0 + (u32)g : wline0;    // dead lines work on the wave's first line (no stores)
    const u32 cA = (u32)a.cntA;
    const u32 bq = gidx / cA, qq = gidx - bq * cA;
    const u32 jP = 1u + (u32)a.cP + 2u * qq, jQ = 1u + (u32)a.cQ + 2u * bq;
    const u32 slot = a.rs.slot0 + gidx;
    const u32 per = (u32)a.seg;                   // block slots per line and entry of the factor
    const u32 csL = a.rs.csL, csP = a.rs.csP, csQ = a.rs.csQ;
    const double ihP[2] = {a.rs.ihP[jP - 1], a.rs.ihP[jP]};
    const double ihQ[2] = {a.rs.ihQ[jQ - 1], a.rs.ihQ[jQ]};
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    const int type = (r == 0) ? 0 : (r <= 2 ? 1 : 2);
    const bool t0 = type == 0;
    const int side = t0 ? 0 : ((r - 1) & 1);
    const double sg = side ? -1.0 : 1.0;
    // field offsets (elements) of the row's own unknown [0] and its six neighbour values [1..6] at block 0, strides per block
#define FL_(vL, vP, vQ) (a.rs.off[0] + (vL) * a.rs.st[0][0] + (vP) * a.rs.st[0][1] + (vQ) * a.rs.st[0][2])
#define FP_(vL, vP, vQ) (a.rs.off[1] + (vL) * a.rs.st[1][0] + (vP) * a.rs.st[1][1] + (vQ) * a.rs.st[1][2])
#define FQ_(vL, vP, vQ) (a.rs.off[2] + (vL) * a.rs.st[2][0] + (vP) * a.rs.st[2][1] + (vQ) * a.rs.st[2][2])
    u32 ob[7], os[7];
    double K[6];
    if (type == 0) {
        ob[0] = FL_(0u, jP, jQ);
        ob[1] = FL_(0u, jP + 1u, jQ); ob[2] = FL_(0u, jP - 1u, jQ); ob[3] = FL_(0u, jP, jQ + 1u); ob[4] = FL_(0u, jP, jQ - 1u);
        ob[5] = ob[1]; ob[6] = ob[1];
#pragma unroll
        for (int t = 0; t < 7; ++t) os[t] = a.rs.st[0][0];
        K[0] = kP[1] * ihP[1]; K[1] = kP[0] * ihP[0]; K[2] = kQ[1] * ihQ[1]; K[3] = kQ[0] * ihQ[0];
        K[4] = 0.0; K[5] = 0.0;
    } else if (type == 1) {
        const u32 pcell = jP - 1u + (u32)side, pnode = side ? jP + 1u : jP - 1u;
        ob[0] = FP_(1u, pcell, jQ);
        ob[1] = FL_(1u, pnode, jQ); ob[2] = FL_(0u, pnode, jQ);
        ob[3] = FQ_(1u, pnode, jQ); ob[4] = FQ_(1u, pnode, jQ - 1u);
        ob[5] = FP_(1u, pcell, jQ + 1u); ob[6] = FP_(1u, pcell, jQ - 1u);
        os[0] = a.rs.st[1][0]; os[1] = a.rs.st[0][0]; os[2] = a.rs.st[0][0];
        os[3] = a.rs.st[2][0]; os[4] = a.rs.st[2][0]; os[5] = a.rs.st[1][0]; os[6] = a.rs.st[1][0];
        const double ihA = side ? ihP[1] : ihP[0];
        K[0] = sg * ihA; K[1] = -sg * ihA;                      // x kL[1], x kL[0] per block
        K[2] = sg * kQ[1] * ihA; K[3] = -sg * kQ[0] * ihA;
        K[4] = kQ[1] * ihQ[1]; K[5] = kQ[0] * ihQ[0];
    } else {
        const u32 qcell = jQ - 1u + (u32)side, qnode = side ? jQ + 1u : jQ - 1u;
        ob[0] = FQ_(1u, jP, qcell);
        ob[1] = FL_(1u, jP, qnode); ob[2] = FL_(0u, jP, qnode);
        ob[3] = FP_(1u, jP, qnode); ob[4] = FP_(1u, jP - 1u, qnode);
        ob[5] = FQ_(1u, jP + 1u, qcell); ob[6] = FQ_(1u, jP - 1u, qcell);
        os[0] = a.rs.st[2][0]; os[1] = a.rs.st[0][0]; os[2] = a.rs.st[0][0];
        os[3] = a.rs.st[1][0]; os[4] = a.rs.st[1][0]; os[5] = a.rs.st[2][0]; os[6] = a.rs.st[2][0];
        const double ihA = side ? ihQ[1] : ihQ[0];
        K[0] = sg * ihA; K[1] = -sg * ihA;
        K[2] = sg * kP[1] * ihA; K[3] = -sg * kP[0] * ihA;
        K[4] = kP[1] * ihP[1]; K[5] = kP[0] * ihP[0];
    }
#undef FL_
#undef FP_
#undef FQ_
    const T* __restrict__ e = (a.e + boff_);
    const T* __restrict__ s = (a.s + boff_);
    const T* wline = a.fac + (i64)slot * (15 * per);
    u32 cface0 = (jP - 1u) * csP + (jQ - 1u) * csQ;
    // the row's own zeta pair at the next cell: rows 1,2 (P side; Q 0,1), rows 3,4 (P 0,1; Q side); row 0: not used
    const u32 pa = (type == 1) ? (u32)side * csP : (type == 2) ? (u32)side * csQ : 0u;
    const u32 pb = (type == 1) ? csQ : (type == 2) ? csP : 0u;
    u32 wre[5];                 // entries of the row's own factor row
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
        const int e0 = wpk(0, cc), e1 = wpk(1, cc), e2 = wpk(2, cc), e3 = wpk(3, cc), e4 = wpk(4, cc);
        wre[cc] = (u32)((r == 0) ? e0 : (r == 1) ? e1 : (r == 2) ? e2 : (r == 3) ? e3 : e4) * per;
    }
    const int qslot = t0 ? 4 : r - 1;            // the row's lane in chain mode
    if (dbg & 8) {       // (timing only, with 4: the stores would go astray)
        wline = a.fac; cface0 = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) { ob[t] = a.rs.off[0] + a.rs.st[0][0] + a.rs.st[0][1] + a.rs.st[0][2]; os[t] = 0; }
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) wre[cc] = 0;
    }

    // All loads are unconditional (block index clamped into the line): no branch between a load and its use, so the
    // compiler keeps the prefetched sets in flight with counted waits.  rd = tick (36 / NL blocks of each line).
    auto load_fwd = [&](PcFwd<T>& d, int rd) __attribute__((always_inline)) {
        const int i = rd * C + bsub;
        const int ic = (dbg & 8) ? 0 : (i < nL ? i : nL - 1);
        const bool lastb = (ic == nL - 1);
        const T* w = wline + ic;
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) d.W[cc] = w[wre[cc]];
        const u32 cface = cface0 + (u32)ic * csL;
        d.f[0] = a.zeta[cface]; d.f[1] = a.zeta[cface + csP]; d.f[2] = a.zeta[cface + csQ]; d.f[3] = a.zeta[cface + csP + csQ];
        const u32 cnext = lastb ? 0u : csL;
        d.n0 = a.zeta[cface + cnext + pa]; d.n1 = a.zeta[cface + cnext + pa + pb];
        d.ihl0 = a.rs.ihL[ic]; d.ihl1 = a.rs.ihL[lastb ? ic : ic + 1];
        // (the last block has no transverse rows: their loads are clamped to the block before, values unused)
        const u32 ie = (u32)((!t0 && lastb) ? (ic > 0 ? ic - 1 : 0) : ic);
#pragma unroll
        for (int t = 0; t < 6; ++t) d.E[t] = e[ob[1 + t] + ie * os[1 + t]];
        d.S = s[ob[0] + ie * os[0]];
    };
    // coupling coefficients of A_i (core.py:684-691) from the zeta face at cell i: row 0 = a_k, diagonal = d_k
    auto left_coef = [&](int i, const double (&f)[4], double ihl0, double (&av)[4], double (&dv)[4]) {
        const double pP0 = f[0] + f[2], pP1 = f[1] + f[3], pQ0 = f[0] + f[1], pQ1 = f[2] + f[3];
        const double act = (i > 0 && i < nL) ? 1.0 : 0.0;
        const double t1 = act * ihl0, t2 = -0.5 * t1 * ihl0;
        av[0] = kP[0] * pP0 * t1; av[1] = -kP[1] * pP1 * t1; av[2] = kQ[0] * pQ0 * t1; av[3] = -kQ[1] * pQ1 * t1;
        dv[0] = t2 * pP0; dv[1] = t2 * pP1; dv[2] = t2 * pQ0; dv[3] = t2 * pQ1;
    };


    if (wv == 0) {
        // =========================================== the chain wave ===========================================
        // chain row cq of line cg = lane cq of DPP row cg.  Forward: cq 0..3 = u (block rows 1..4), 4 = z[0]; backward:
        // cq 0..3 = v, 4..8 = x[0..4].  The remaining lanes shadow lane cq % 5 resp. cq % 9 (same reads, same arithmetic,
        // their LDS writes go to a private dump slot): no divergence around the DPP instructions.
        const int crow = lane >> 4, cq = lane & 15;
        const int cg = crow % NL;
        T* const dump = dumpb + lane;
        T u = Zero<T>::v();
        for (int t = 0; t <= nch; ++t) {
            if (t > 0) {
                const int c = t - 1;
                const int ns = (dbg & 1) ? 0 : min(C, nL - c * C);
                const bool act = crow < NL && cq < 5;
                const T* q = stage + ((c & 1) * 36 * 9 + cg * 9 + (cq % 5)) * 5;
                T* pk = act ? park + ((c * C * NL + cg) * 5 + cq) : dump;
                const int pstep = act ? NL * 5 : 0;
                // Operands of step bc + 1 are read while step bc computes; two operand sets in alternation (loop unrolled by
                // two), so that the accumulator of a step is the register its constant term was read into.
                T cX = q[0], GX[4] = {q[1], q[2], q[3], q[4]}, cY, GY[4];
                int bc = 0;
                for (; bc + 2 <= ns; bc += 2) {
                    q += NL * 45;
                    cY = q[0]; GY[0] = q[1]; GY[1] = q[2]; GY[2] = q[3]; GY[3] = q[4];
                    __builtin_amdgcn_sched_barrier(0);
                    pc_chain_step(cX, u, GX);
                    pk[0] = cX;
                    __builtin_amdgcn_sched_barrier(0);
                    if (bc + 2 < ns) q += NL * 45;
                    u = cX;               // (lanes >= 4 of a row: never read by the broadcasts)
                    cX = q[0]; GX[0] = q[1]; GX[1] = q[2]; GX[2] = q[3]; GX[3] = q[4];
                    __builtin_amdgcn_sched_barrier(0);
                    pc_chain_step(cY, u, GY);
                    pk[pstep] = cY;
                    u = cY;
                    pk += 2 * pstep;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (bc < ns) {
                    pc_chain_step(cX, u, GX);
                    pk[0] = cX;
                    u = cX;
                }
            }
            __syncthreads();
        }
        T v = Zero<T>::v();
        for (int t = 0; t <= nch + 1; ++t) {
            if (t >= 1 && t <= nch) {
                const int c = nch - t;
                const int ns = (dbg & 1) ? 0 : min(C, nL - c * C);
                const bool act = crow < NL && cq >= 4 && cq < 9;
                const T* q = stage + ((((t - 1) & 1) * 36 + (ns > 0 ? ns - 1 : 0) * NL + cg) * 9 + (cq % 9)) * 5;
                T* pk = act ? const_cast<T*>(q) : dump;             // x replaces the row's constant term
                const int pstep = act ? NL * 45 : 0;
                T cX = q[0], GX[4] = {q[1], q[2], q[3], q[4]}, cY, GY[4];
                int bc = ns;
                for (; bc >= 2; bc -= 2) {
                    q -= NL * 45;
                    cY = q[0]; GY[0] = q[1]; GY[1] = q[2]; GY[2] = q[3]; GY[3] = q[4];
                    __builtin_amdgcn_sched_barrier(0);
                    pc_chain_step(cX, v, GX);
                    pk[0] = cX;
                    __builtin_amdgcn_sched_barrier(0);
                    if (bc > 2) q -= NL * 45;
                    v = cX;               // (lanes >= 4 of a row: never read by the broadcasts)
                    cX = q[0]; GX[0] = q[1]; GX[1] = q[2]; GX[2] = q[3]; GX[3] = q[4];
                    __builtin_amdgcn_sched_barrier(0);
                    pc_chain_step(cY, v, GY);
                    pk[-pstep] = cY;
                    v = cY;
                    pk -= 2 * pstep;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (bc > 0) {
                    pc_chain_step(cX, v, GX);
                    pk[0] = cX;
                    v = cX;
                }
            }
            __syncthreads();
        }
        return;
    }

    // ============================================= the producer waves =============================================
    T* const xr = xb + pw * 60;
    auto produce_fwd = [&](const PcFwd<T>& d, int rd) __attribute__((always_inline)) {
        const int i = rd * C + bsub;
        const bool lastb = i >= nL - 1;
        // the row's right-hand side (core.py:697-736, regrouped per row as in k_line_sweep_rp)
        // (values first, selects second: a conditional between two members of the set would be compiled as a runtime index
        // into it, which moves the whole register set to scratch memory)
        const double f00 = d.f[0], f10 = d.f[1], f01 = d.f[2], f11 = d.f[3], dn0 = d.n0, dn1 = d.n1;
        const double z0 = t0 ? f00 : (type == 1) ? (side ? f10 : f00) : (side ? f01 : f00);
        const double z1 = t0 ? f01 : (type == 1) ? (side ? f11 : f01) : (side ? f11 : f10);
        const double n0 = t0 ? f10 : dn0, n1 = t0 ? f11 : dn1;
        const double kL0 = 0.5 * d.ihl0, kL1 = 0.5 * d.ihl1;
        const double rs0 = z0 + z1, rs1 = n0 + n1, cs0 = z0 + n0, cs1 = z1 + n1;
        T y = d.S;
        cmac(y, d.E[0], (t0 ? K[0] : K[0] * kL1) * rs1);
        cmac(y, d.E[1], (t0 ? K[1] : K[1] * kL0) * rs0);
        cmac(y, d.E[2], K[2] * cs1);
        cmac(y, d.E[3], K[3] * cs0);
        cmac(y, d.E[4], K[4] * cs1);
        cmac(y, d.E[5], K[5] * cs0);
        if (!t0 && lastb) y = Zero<T>::v();
        const int il = pl - 12 * r;             // the block's slot among the wave's twelve
        if (pact) xr[il * 5 + r] = y;
        double av[4], dv[4];
        left_coef(i, d.f, d.ihl0, av, dv);
        // row r of the block map: c_r = (W b)[r], G_rk = -(W[r][0] a_k + W[r][k+1] d_k)   (blocks beyond the line: never read)
        T mG[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T gk = Zero<T>::v();
            cmsc(gk, d.W[0], av[k]);
            cmsc(gk, d.W[k + 1], dv[k]);
            mG[k] = gk;
        }
        __builtin_amdgcn_wave_barrier();
        const T* xi = xr + il * 5;
        T mc = d.W[0] * xi[0];
#pragma unroll
        for (int cc = 1; cc < 5; ++cc) cmac(mc, d.W[cc], xi[cc]);
        if (pact) {
            T* st = stage + (((rd & 1) * 36 + it) * 9 + qslot) * 5;
            st[0] = mc;
#pragma unroll
            for (int k = 0; k < 4; ++k) st[1 + k] = mG[k];
        }
        __builtin_amdgcn_wave_barrier();
    };
    {
        PcFwd<T> A, B, Cs;
        load_fwd(A, 0); load_fwd(B, 1); load_fwd(Cs, 2);
        int t = 0;
        for (; t + 3 <= nch + 1; t += 3) {
            produce_fwd(A, t); load_fwd(A, t + 3); __syncthreads();
            produce_fwd(B, t + 1); load_fwd(B, t + 4); __syncthreads();
            produce_fwd(Cs, t + 2); load_fwd(Cs, t + 5); __syncthreads();
        }
        if (t < nch + 1) {
            produce_fwd(A, t); __syncthreads();
            if (t + 1 < nch + 1) { produce_fwd(B, t + 1); __syncthreads(); }
        }
    }

    T* __restrict__ eo = (a.e + boff_);
    auto load_bwd = [&](PcBwd<T>& d, int rd) __attribute__((always_inline)) {
        const int i = (rd > 0 ? rd : 0) * C + bsub;
        const int ic = (dbg & 8) ? 0 : (i < nL ? i : nL - 1);
        const T* w = wline + ic;
#pragma unroll
        for (int l = 0; l < 4; ++l) d.Wt[l] = w[wre[l + 1]];
#pragma unroll
        for (int l = 0; l < 4; ++l) d.W0t[l] = w[(u32)(wpk(0, l + 1)) * per];
        const u32 cface = cface0 + (u32)ic * csL;
        d.f[0] = a.zeta[cface]; d.f[1] = a.zeta[cface + csP]; d.f[2] = a.zeta[cface + csQ]; d.f[3] = a.zeta[cface + csP + csQ];
        d.ihl0 = a.rs.ihL[ic];
    };
    // tick tt of the way back: store the x of the chunk that was produced two ticks ago (same stage buffer, same slots), then
    // produce chunk rd = nch - 1 - tt (rd < 0: nothing left; the values written are never read)
    auto produce_bwd = [&](const PcBwd<T>& d, int tt) __attribute__((always_inline)) {
        T* st = stage + ((tt & 1) * 36 + it) * 45;
        if (tt >= 2) {
            const int i = (nch + 1 - tt) * C + bsub;
            if (pact && live && i < nL && (t0 || i != nL - 1) && !(dbg & 4)) eo[ob[0] + (u32)i * os[0]] = st[(4 + r) * 5];
        }
        const int rd = nch - 1 - tt;
        const int i = (rd > 0 ? rd : 0) * C + bsub;
        const int ic = i < nL ? i : nL - 1;
        double av[4], dv[4];
        left_coef(i, d.f, d.ihl0, av, dv);
        const T* pz = park + (ic * NL + g) * 5;
        const T z0 = pz[4];
        const T zr = pz[qslot];                 // the row's own z (row 0: z[0])
        __builtin_amdgcn_wave_barrier();
        if (pact) {
            // chain rows 4 + r: x_r = z_r - sum_l W[r][l+1] v_{i+1,l}
            T* sx = st + (4 + r) * 5;
            sx[0] = zr;
#pragma unroll
            for (int l = 0; l < 4; ++l) sx[1 + l] = -d.Wt[l];
            if (!t0) {
                // chain row r - 1 of the backward map: g = a z_0 + d z_r,  H_l = -(a W[0][l+1] + d W[r][l+1])
                const double ar = (r == 1) ? av[0] : (r == 2) ? av[1] : (r == 3) ? av[2] : av[3];
                const double dr = (r == 1) ? dv[0] : (r == 2) ? dv[1] : (r == 3) ? dv[2] : dv[3];
                T gg = z0 * ar;
                cmac(gg, zr, dr);
                T* sv = st + (r - 1) * 5;
                sv[0] = gg;
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    T h = Zero<T>::v();
                    cmsc(h, d.W0t[l], ar);
                    cmsc(h, d.Wt[l], dr);
                    sv[1 + l] = h;
                }
            }
        }
    };
    {
        PcBwd<T> A, B, Cs;
        load_bwd(A, nch - 1); load_bwd(B, nch - 2); load_bwd(Cs, nch - 3);
        int t = 0;
        for (; t + 3 <= nch + 2; t += 3) {
            produce_bwd(A, t); load_bwd(A, nch - 4 - t); __syncthreads();
            produce_bwd(B, t + 1); load_bwd(B, nch - 5 - t); __syncthreads();
            produce_bwd(Cs, t + 2); load_bwd(Cs, nch - 6 - t); __syncthreads();
        }
        if (t < nch + 2) {
            produce_bwd(A, t); __syncthreads();
            if (t + 1 < nch + 2) { produce_bwd(B, t + 1); __syncthreads(); }
        }
    }
}
