// Quad-per-block line sweep: parallel IN the line (reference emg3d/core.py:477-1316 line solves,
// same cached block factorisation as smooth.hpp).
//
// The two recurrences of a line solve
//   forward : z_i = W_i (b_i - A_i z_{i-1})          backward: x_i = z_i - W_i A_{i+1}^T x_{i+1}
// couple neighbouring blocks only through the four transverse unknowns (A_i has a zero first
// column), so each is a chain of affine maps of C^4,
//   forward : u_i = c_i + G_i u_{i-1},  u = z[1..4],        G_i = -(W_i A_i)[1..4][1..4]
//   backward: v_i = g_i + H_i v_{i+1},  v = A_i^T x_i,       H_i = -(A_i^T W_i)[1..4][1..4]  (= G_i^T)
// and a chain of affine maps is a prefix scan.  Here FOUR lanes (a quad) own one 5x5 block: lane r
// holds ROW r of the block's map (c_r and G_r., five numbers) and composing two maps needs only the
// own row plus the whole other map -- new row = own row o other -- which the other quad publishes in
// LDS.  A line of nL blocks is a segment of SEG = 2^k >= nL quads; a Kogge-Stone scan over the
// segment takes log2(SEG) steps of 20 complex MACs per lane instead of nL/2 dependent block steps.
// Short lines share a wave (16 / SEG lines), long lines span the waves of a workgroup.
//
// A quad may own M = 2 consecutive blocks: their maps are composed inside the quad first (through
// the quad's own LDS slot), which halves the scan work per block at the price of registers.
//
// Per lane and block: row r+1 and row 0 of the cached inverse W_i (read ONCE, kept in registers for both
// passes), the right-hand side of row r+1 and one of the four terms of row 0 (summed over the quad
// with DPP), the coupling coefficients of A_i (the 2x2 zeta face at cell i).  Lane r stores the
// transverse unknown r+1, lane 0 also the unknown along the line.
//
// Factor layout (k_line_factor with LineArgs::qpl set, one-sided): [line][entry][M * SEG block slots].
#pragma once
#include "smooth.hpp"


// broadcast lane K of every quad (DPP quad_perm, no LDS)
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ c128 quad_bcast(c128 v) { return mk(quad_bcast<K>(v.re), quad_bcast<K>(v.im)); }
// sum over the quad (butterfly: xor 1, xor 2)
__device__ __forceinline__ double quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, false),     // quad_perm [1,0,3,2]
                                __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, false));
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4e, 0xf, 0xf, false),            // quad_perm [2,3,0,1]
                         __builtin_amdgcn_update_dpp(0, lo, 0x4e, 0xf, 0xf, false));
    return v + o;
}
__device__ __forceinline__ c128 quad_sum(c128 v) { return mk(quad_sum(v.re), quad_sum(v.im)); }

// value of the lane four below (UP) / four above: inside a row of 16 lanes by DPP row_shr:4 / row_shl:4 (lanes without a source get 0) ...
template <bool UP>
__device__ __forceinline__ double row_shift4(double v) {
    constexpr int ctrl = UP ? 0x114 : 0x104;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool UP> __device__ __forceinline__ c128 row_shift4(c128 v) { return mk(row_shift4<UP>(v.re), row_shift4<UP>(v.im)); }
// ... across the wave by lane shuffles (8-block lines span two rows)
template <bool UP> __device__ __forceinline__ double lane_shift4(double v) { return UP ? __shfl_up(v, 4) : __shfl_down(v, 4); }
template <bool UP> __device__ __forceinline__ c128 lane_shift4(c128 v) { return mk(lane_shift4<UP>(v.re), lane_shift4<UP>(v.im)); }

// Every member of LineArgs this kernel reads, loaded in one burst (EMG_ARGS_BURST, common.hpp: a launch on a coarse level lives
// for 5-8 us, and its prologue used to be a chain of ~10 dependent scalar round trips through the 592-byte argument struct).
template <class T>
__device__ __forceinline__ void qpl_args_burst(const LineArgs<T>& a) {
    asm volatile("" :: "s"(a.e), "s"(a.s), "s"(a.fac), "s"(a.zeta), "s"(a.rs.ihL), "s"(a.rs.ihP), "s"(a.rs.ihQ), "s"(a.bt.st),
                 "s"(a.bt.mask), "s"(a.bt.n), "s"(a.xcd), "s"(a.mode), "s"(a.cntA), "s"(a.cntB), "s"(a.cP), "s"(a.cQ), "s"(a.seg),
                 "s"(a.rs.nL), "s"(a.rs.csL), "s"(a.rs.csP), "s"(a.rs.csQ), "s"(a.rs.slot0), "s"(a.rs.off[0]), "s"(a.rs.off[1]),
                 "s"(a.rs.off[2]));
    asm volatile("" :: "s"(a.rs.st[0][0]), "s"(a.rs.st[0][1]), "s"(a.rs.st[0][2]), "s"(a.rs.st[1][0]), "s"(a.rs.st[1][1]),
                 "s"(a.rs.st[1][2]), "s"(a.rs.st[2][0]), "s"(a.rs.st[2][1]), "s"(a.rs.st[2][2]), "s"(a.t), "s"(a.jQ0), "s"(a.cnt),
                 "s"(a.rs.nP), "s"(a.rs.nQ), "s"(a.qd), "s"(a.qdn));
}

// DM (descriptor mode, colour order only): 0 = the prologue computes indices and coefficients; 1 = it computes them, WRITES them to
// LineArgs::qd and returns (run once per (level, direction, colour) when the factor is built); 2 = it loads them.  Of the 3.6 us a
// launch on a level of <= 16-block lines spends inside the kernel, 0.8 are index arithmetic and 0.15 coefficient products that are the
// same in every one of the 420 such launches of a 128^3 F-cycle (profiles/r05_qpl_stamps.txt, HISTORY R5.5 / R5.12).
// CH (round 6): on lines of <= 8 blocks the two Kogge-Stone scans are replaced by CHAINS across the quads of the line -- quad q
// takes z[1..4] of quad q - 1 from four lanes below (DPP row shift on 4-block lines, lane shuffles on 8-block lines), broadcasts the
// four numbers inside the quad and applies its own row of the block map (4 complex multiply-adds), seg - 1 times; no LDS, no
// barrier, 20 instead of 80 FP64 instructions per step.  Every quad runs every step (its value is final after step q and stays).
template <class T, int NW, int M, bool HL = false, int DM = 0, bool CH = false>      // HL: hyperplane loop (mode 2, lexicographic order)
__global__ __launch_bounds__(64 * NW) void k_line_sweep_qpl(LineArgs<T> a) {
    static_assert(DM == 0 || (!HL && NW == 1), "descriptors: one wave per workgroup, colour order");
    static_assert(!CH || (NW == 1 && M == 1 && !HL), "chain form: lines inside one wave, one block per quad");
    constexpr int NQ = 16 * NW;                 // quads per workgroup; a quad owns M consecutive blocks
#ifdef EMG3D_LAB
    // lab: cycle-counter stamps of workgroup 0 (EMG3D_Q_TILE=512): entry, arguments in, loads issued, loads in, forward scan done,
    // backward scan done, stores issued
    long long qts[7] = {0, 0, 0, 0, 0, 0, 0};
#define QPL_TS(i) do { if (a.tile & 512) { asm volatile("" ::: "memory"); qts[i] = (long long)__builtin_readcyclecounter(); asm volatile("" ::: "memory"); } } while (0)
#else
#define QPL_TS(i) do {} while (0)
#endif
    QPL_TS(0);
    qpl_args_burst(a);
    QPL_TS(1);
    const int tid = threadIdx.x;
    const int quad = tid >> 2, r = tid & 3;
    const int seg = a.seg;                      // quads per line (power of two, M * seg >= nL)
    const int ch = quad & (seg - 1);            // chunk of the line
    const int lseg = __builtin_ctz((unsigned)seg);      // (a power of two: shifts instead of the division sequences)
    const int lpg = NQ >> lseg;                 // lines per workgroup
    // All index arithmetic in 32 bits: the host admits this kernel only when every array is shorter than
    // 2^32 bytes (MG::rp_fits), so element offsets and line counts fit comfortably.
    typedef unsigned int u32;
    EMG_SWEEP_WG(a)
    // ---- exchange buffer: per quad the four rows of its map, five numbers each; double buffered --
    __shared__ T xb[2][NQ][4][5];
    // One pass over the lines gline = 0 .. nlines-1 of a colour (mode 0) or of the hyperplane jP + 2 jQ = t_ (jQ from jQ0_)
    auto one = [&](const u32 gline, const u32 nlines, const u32 t_, const u32 jQ0_) {
    const bool live = gline < nlines;
    const int nL = (int)a.rs.nL;
    // Loads as scalar base + 32-bit byte offset (every array is shorter than 2^32 bytes): the typed form base[u32 index] costs a
    // 64-bit address pair per load
    auto ld_d = [](const double* base, u32 idx) -> double { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + idx * 8u); };
    auto ld_t = [](const T* base, u32 idx) -> T { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (u32)sizeof(T)); };
    const T* __restrict__ e = (a.e + boff_);
    const T* __restrict__ s = (a.s + boff_);
    // ---- what the scans and the stores take from the prologue ----
    T Wr[M][5], W0[M][5];       // rows r+1 and 0 of the cached inverse
    T b[M][5];                  // right-hand side (all five rows, in every lane of the quad)
    double av[M][4], dv[M][4];  // A_i: row 0 = a_k, diagonal = d_k
    bool lastb[M], inl[M];
    u32 dS[M], dS0[M];          // element offsets of the row's own edge and of the edge along the line (source load = result store)
    // descriptor of this thread (DM != 0): [item][LineArgs::qdn threads], thread = logical workgroup x 64 + lane
    const u32 tix = (u32)wg * (64u * NW) + (u32)tid;
    constexpr int QD_U4 = 3, QD_D2 = 8;         // per block: 3 x uint4 (11 offsets / flags) + 8 x double2 (15 coefficient products)
    if constexpr (DM == 2) {
        // ================= descriptors loaded: offsets and coefficient products come from the table =================
        const uint4* const q4 = reinterpret_cast<const uint4*>(a.qd);
        const emg_d2* const c2 = reinterpret_cast<const emg_d2*>(q4 + (size_t)(QD_U4 * M) * a.qdn);
        uint4 u[M][QD_U4];
        emg_d2 cf[M][QD_D2];
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int k = 0; k < QD_U4; ++k) u[j][k] = q4[(u32)(QD_U4 * j + k) * a.qdn + tix];
#pragma unroll
            for (int k = 0; k < QD_D2; ++k) cf[j][k] = c2[(u32)(QD_D2 * j + k) * a.qdn + tix];
        }
        QPL_TS(2);
        T E[M][6], S[M], E0[M], S0[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const u32 oE[6] = {u[j][0].x, u[j][0].y, u[j][0].z, u[j][0].w, u[j][1].x, u[j][1].y};
            dS[j] = u[j][1].z; dS0[j] = u[j][2].x;
            inl[j] = (u[j][2].z & 1u) != 0; lastb[j] = (u[j][2].z & 2u) != 0;
            const T* w = a.fac + u[j][2].y;
#pragma unroll
            for (int c = 0; c < 5; ++c) W0[j][c] = w[(u32)(wpk(0, c) * (M * seg))];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int e1 = wpk(1, c), e2 = wpk(2, c), e3 = wpk(3, c), e4 = wpk(4, c);
                const int en = (r == 0) ? e1 : (r == 1) ? e2 : (r == 2) ? e3 : e4;
                Wr[j][c] = w[(u32)(en * (M * seg))];
            }
#pragma unroll
            for (int t = 0; t < 6; ++t) E[j][t] = ld_t(e, oE[t]);
            S[j] = ld_t(s, dS[j]);
            E0[j] = ld_t(e, u[j][1].w);
            S0[j] = ld_t(s, dS0[j]);
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double cE[6] = {cf[j][0].x, cf[j][0].y, cf[j][1].x, cf[j][1].y, cf[j][2].x, cf[j][2].y};
            av[j][0] = cf[j][3].y; av[j][1] = cf[j][4].x; av[j][2] = cf[j][4].y; av[j][3] = cf[j][5].x;
            dv[j][0] = cf[j][5].y; dv[j][1] = cf[j][6].x; dv[j][2] = cf[j][6].y; dv[j][3] = cf[j][7].x;
            T y = S[j];
#pragma unroll
            for (int t = 0; t < 6; ++t) cmac(y, E[j][t], cE[t]);
            const T bo = lastb[j] ? Zero<T>::v() : y;
            T part = E0[j] * cf[j][3].x;
            part = quad_sum(part);
            b[j][0] = S0[j] + part;
            b[j][1] = quad_bcast<0>(bo); b[j][2] = quad_bcast<1>(bo); b[j][3] = quad_bcast<2>(bo); b[j][4] = quad_bcast<3>(bo);
            if (!inl[j]) {      // beyond the line: the zero map
#pragma unroll
                for (int c = 0; c < 5; ++c) { Wr[j][c] = Zero<T>::v(); W0[j][c] = Zero<T>::v(); }
            }
        }
    } else {
    // ================= indices and coefficients computed here (DM == 1: ... written to the table, nothing else) =================
    const u32 gidx = live ? gline : 0u;         // dead lines work on line 0 (no stores): barriers stay uniform
    u32 jP, jQ;
    if (a.mode == 0) {
        const u32 cA = (u32)a.cntA;
        const u32 bq = gidx / cA, qq = gidx - bq * cA;
        jP = 1u + (u32)a.cP + 2u * qq;
        jQ = 1u + (u32)a.cQ + 2u * bq;
    } else {
        jQ = jQ0_ + gidx;
        jP = t_ - 2u * jQ;
    }
    // line slot of the factor cache: colour mode numbers the lines of a colour consecutively (slot = first
    // slot of the colour + line index: no per-lane table look-up); hyperplanes mix the colours
    const u32 slot = (a.mode == 0) ? a.rs.slot0 + gidx : (u32)line_slot(a, (i64)jP, (i64)jQ);
    const u32 csL = a.rs.csL, csP = a.rs.csP, csQ = a.rs.csQ;
    const double ihP[2] = {ld_d(a.rs.ihP, jP - 1u), ld_d(a.rs.ihP, jP)};
    const double ihQ[2] = {ld_d(a.rs.ihQ, jQ - 1u), ld_d(a.rs.ihQ, jQ)};
    // ---- row r+1 of a block: a transverse edge at node i+1 (rows 1,2: P-directed at jP-1 / jP;
    //      rows 3,4: Q-directed at jQ-1 / jQ).  Same regrouping of the reference's m-coefficients
    //      (core.py:609-632, 697-736) as k_line_sweep_tw, written once for "the row's transverse axis
    //      A and the other one B" (A, B = P, Q for rows 1,2 and Q, P for rows 3,4) so that the four
    //      lanes of a quad run the same instructions. ------------------------------------------------
    const bool tp = r < 2;                      // rows 1,2: A = P
    const int side = r & 1;
    const double sg = side ? -1.0 : 1.0;
    const u32 jA = tp ? jP : jQ, jB = tp ? jQ : jP;
    const u32 acell = jA - 1u + (u32)side;       // cell index of the row's edge along A
    const u32 anode = side ? jA + 1u : jA - 1u;  // the neighbouring node line along A
    // component offsets and strides along (L, A, B) for the three components L, A, B
    // (component / axis index: 0 = L, 1 = P, 2 = Q)
    const u32 oLc = a.rs.off[0], oAc = tp ? a.rs.off[1] : a.rs.off[2], oBc = tp ? a.rs.off[2] : a.rs.off[1];
    const u32 sLL = a.rs.st[0][0];
    const u32 sLA = tp ? a.rs.st[0][1] : a.rs.st[0][2], sLB = tp ? a.rs.st[0][2] : a.rs.st[0][1];
    const u32 sAL = tp ? a.rs.st[1][0] : a.rs.st[2][0];
    const u32 sAA = tp ? a.rs.st[1][1] : a.rs.st[2][2], sAB = tp ? a.rs.st[1][2] : a.rs.st[2][1];
    const u32 sBL = tp ? a.rs.st[2][0] : a.rs.st[1][0];
    const u32 sBA = tp ? a.rs.st[2][1] : a.rs.st[1][2], sBB = tp ? a.rs.st[2][2] : a.rs.st[1][1];
    u32 ob[7], os[7];
    ob[0] = oAc + sAL + acell * sAA + jB * sAB;                 // the row's own edge (component A, node i+1)
    ob[1] = oLc + sLL + anode * sLA + jB * sLB;                 // L-edges i+1 and i of the neighbouring line
    ob[2] = ob[1] - sLL;
    ob[3] = oBc + sBL + anode * sBA + jB * sBB;                 // B-edges at the neighbouring node line
    ob[4] = ob[3] - sBB;
    ob[5] = ob[0] + sAB;                                        // A-edges of the B-neighbours
    ob[6] = ob[0] - sAB;
    os[0] = sAL; os[1] = sLL; os[2] = sLL; os[3] = sBL; os[4] = sBL; os[5] = sAL; os[6] = sAL;
    // ---- row 0 (the edge along the line): lane r evaluates term r of its right-hand side ----
    const u32 sLP = a.rs.st[0][1], sLQ = a.rs.st[0][2];
    const u32 o0 = oLc + jP * sLP + jQ * sLQ;
    const u32 db0 = (r < 2) ? sLP : sLQ;          // (two selects; the four-way conditional became divergent branches)
    const u32 ob0 = (r & 1) ? o0 - db0 : o0 + db0;
    const int type = tp ? 1 : 2;

    // ---- per block j of the chunk: ALL loads first (they depend on indices only; the 1/h values requested above are used
    //      only afterwards, so that the launch pays ONE memory round trip for widths, factor, model and fields together), then
    //      coefficients and right-hand side ------------------------------------------------------------------------------
    T E[M][6], S[M], E0[M], S0[M];
    u32 dE[M][6], dE0[M], dF[M];
    double f00[M], f10[M], f01[M], f11[M], n0[M], n1[M], ihl0[M], ihl1[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int i = ch * M + j;                             // block of the line
        const int ic = i < nL ? i : nL - 1;                   // clamped: loads stay in range
        inl[j] = i < nL;
        lastb[j] = (ic == nL - 1);
        if constexpr (DM == 0) {
            // factor layout [line][entry][M * seg block slots]
            // (the factor of a level may pass 4 GiB where its fields do not: 64-bit line offset)
            const T* w = a.fac + ((i64)slot * (15 * (M * seg)) + ic);
#pragma unroll
            for (int c = 0; c < 5; ++c) W0[j][c] = w[(u32)(wpk(0, c) * (M * seg))];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int e1 = wpk(1, c), e2 = wpk(2, c), e3 = wpk(3, c), e4 = wpk(4, c);
                const int en = (r == 0) ? e1 : (r == 1) ? e2 : (r == 2) ? e3 : e4;
                Wr[j][c] = w[(u32)(en * (M * seg))];
            }
        } else {
            dF[j] = slot * (u32)(15 * (M * seg)) + (u32)ic;   // (the host admits descriptors only where the factor has < 2^32 entries)
        }
        // zeta: 2x2 face at cell i (coupling A_i, rhs of row 0, near pair of row r+1), the row's pair at cell i+1
        const u32 cface = (jP - 1u) * csP + (jQ - 1u) * csQ + (u32)ic * csL;
        f00[j] = ld_d(a.zeta, cface); f10[j] = ld_d(a.zeta, cface + csP); f01[j] = ld_d(a.zeta, cface + csQ); f11[j] = ld_d(a.zeta, cface + csP + csQ);
        const u32 cnext = lastb[j] ? 0u : csL;
        const u32 pa = (type == 1) ? (u32)side * csP : (u32)side * csQ;     // rows 1,2: (P side, Q 0/1); 3,4: (P 0/1, Q side)
        const u32 pb = (type == 1) ? csQ : csP;
        n0[j] = ld_d(a.zeta, cface + cnext + pa); n1[j] = ld_d(a.zeta, cface + cnext + pa + pb);
        ihl0[j] = ld_d(a.rs.ihL, (u32)ic); ihl1[j] = ld_d(a.rs.ihL, (u32)(lastb[j] ? ic : ic + 1));
        // fields: own row (clamped on the last block: its transverse rows do not exist)
        const u32 ie = (u32)(lastb[j] ? (ic > 0 ? ic - 1 : 0) : ic);
#pragma unroll
        for (int t = 0; t < 6; ++t) dE[j][t] = ob[1 + t] + ie * os[1 + t];
        dS[j] = ob[0] + ie * os[0];
        dE0[j] = ob0 + (u32)ic * sLL;
        dS0[j] = o0 + (u32)ic * sLL;
        if constexpr (DM == 0) {
#pragma unroll
            for (int t = 0; t < 6; ++t) E[j][t] = ld_t(e, dE[j][t]);
            S[j] = ld_t(s, dS[j]);
            E0[j] = ld_t(e, dE0[j]);
            S0[j] = ld_t(s, dS0[j]);
        }
    }
    QPL_TS(2);
    const double kP[2] = {0.5 * ihP[0], 0.5 * ihP[1]};
    const double kQ[2] = {0.5 * ihQ[0], 0.5 * ihQ[1]};
    double Kc[6];
    {
        // (selects, not ihP[side]: a runtime index into a local array goes through scratch memory)
        const double ihA = tp ? (side ? ihP[1] : ihP[0]) : (side ? ihQ[1] : ihQ[0]);
        const double kB0 = tp ? kQ[0] : kP[0], kB1 = tp ? kQ[1] : kP[1];
        const double ihB0 = tp ? ihQ[0] : ihP[0], ihB1 = tp ? ihQ[1] : ihP[1];
        Kc[0] = sg * ihA; Kc[1] = -sg * ihA;
        Kc[2] = sg * kB1 * ihA; Kc[3] = -sg * kB0 * ihA;
        Kc[4] = kB1 * ihB1; Kc[5] = kB0 * ihB0;
    }
    const double K0 = (r == 0) ? kP[1] * ihP[1] : (r == 1) ? kP[0] * ihP[0] : (r == 2) ? kQ[1] * ihQ[1] : kQ[0] * ihQ[0];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int i = ch * M + j;
        // coefficients
        const double pP0 = f00[j] + f01[j], pP1 = f10[j] + f11[j], pQ0 = f00[j] + f10[j], pQ1 = f01[j] + f11[j];   // zeta pair sums at cell i
        const double act = (i > 0 && i < nL) ? 1.0 : 0.0;          // A_i (core.py:684-691); none for block 0
        const double t1 = act * ihl0[j], t2 = -0.5 * t1 * ihl0[j];
        av[j][0] = kP[0] * pP0 * t1; av[j][1] = -kP[1] * pP1 * t1; av[j][2] = kQ[0] * pQ0 * t1; av[j][3] = -kQ[1] * pQ1 * t1;
        dv[j][0] = t2 * pP0; dv[j][1] = t2 * pP1; dv[j][2] = t2 * pQ0; dv[j][3] = t2 * pQ1;
        // the six coefficient products of b_{r+1} and the one of this lane's term of b_0
        double cE[6];
        {
            const double z0 = (type == 1) ? (side ? f10[j] : f00[j]) : (side ? f01[j] : f00[j]);
            const double z1 = (type == 1) ? (side ? f11[j] : f01[j]) : (side ? f11[j] : f10[j]);
            const double kL0 = 0.5 * ihl0[j], kL1 = 0.5 * ihl1[j];
            const double rs0 = z0 + z1, rs1 = n0[j] + n1[j];
            const double cs0 = z0 + n0[j], cs1 = z1 + n1[j];
            cE[0] = (Kc[0] * kL1) * rs1; cE[1] = (Kc[1] * kL0) * rs0;
            cE[2] = Kc[2] * cs1; cE[3] = Kc[3] * cs0; cE[4] = Kc[4] * cs1; cE[5] = Kc[5] * cs0;
        }
        const double c0 = K0 * ((r == 0) ? pP1 : (r == 1) ? pP0 : (r == 2) ? pQ1 : pQ0);
        if constexpr (DM == 1) {
            uint4* const q4 = reinterpret_cast<uint4*>(const_cast<void*>(a.qd));
            emg_d2* const c2 = reinterpret_cast<emg_d2*>(q4 + (size_t)(QD_U4 * M) * a.qdn);
            const u32 flags = (inl[j] ? 1u : 0u) | (lastb[j] ? 2u : 0u) | (live ? 4u : 0u);
            q4[(u32)(QD_U4 * j + 0) * a.qdn + tix] = make_uint4(dE[j][0], dE[j][1], dE[j][2], dE[j][3]);
            q4[(u32)(QD_U4 * j + 1) * a.qdn + tix] = make_uint4(dE[j][4], dE[j][5], dS[j], dE0[j]);
            q4[(u32)(QD_U4 * j + 2) * a.qdn + tix] = make_uint4(dS0[j], dF[j], flags, 0u);
            auto d2 = [](double x, double y) { emg_d2 v; v.x = x; v.y = y; return v; };
            c2[(u32)(QD_D2 * j + 0) * a.qdn + tix] = d2(cE[0], cE[1]);
            c2[(u32)(QD_D2 * j + 1) * a.qdn + tix] = d2(cE[2], cE[3]);
            c2[(u32)(QD_D2 * j + 2) * a.qdn + tix] = d2(cE[4], cE[5]);
            c2[(u32)(QD_D2 * j + 3) * a.qdn + tix] = d2(c0, av[j][0]);
            c2[(u32)(QD_D2 * j + 4) * a.qdn + tix] = d2(av[j][1], av[j][2]);
            c2[(u32)(QD_D2 * j + 5) * a.qdn + tix] = d2(av[j][3], dv[j][0]);
            c2[(u32)(QD_D2 * j + 6) * a.qdn + tix] = d2(dv[j][1], dv[j][2]);
            c2[(u32)(QD_D2 * j + 7) * a.qdn + tix] = d2(dv[j][3], 0.0);
        } else {
            T y = S[j];
#pragma unroll
            for (int t = 0; t < 6; ++t) cmac(y, E[j][t], cE[t]);
            const T bo = lastb[j] ? Zero<T>::v() : y;       // b_{r+1} (zero on the last block)
            T part = E0[j] * c0;
            part = quad_sum(part);
            b[j][0] = S0[j] + part;
            b[j][1] = quad_bcast<0>(bo); b[j][2] = quad_bcast<1>(bo); b[j][3] = quad_bcast<2>(bo); b[j][4] = quad_bcast<3>(bo);
            if (!inl[j]) {      // beyond the line: the zero map
#pragma unroll
                for (int c = 0; c < 5; ++c) { Wr[j][c] = Zero<T>::v(); W0[j][c] = Zero<T>::v(); }
            }
        }
    }
    if constexpr (DM == 1) return;      // (generating mode: the table is written, nothing is swept)
    }

    QPL_TS(3);
    // (lines of <= 16 quads live in one wave whatever the workgroup's size: a wave-level barrier suffices)
    auto sync = [&]() { if (NW > 1 && (HL || seg > 16)) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    T mc, mG[4];        // my row of the chunk map: u -> mc + mG . u
    auto publish = [&](int p) {
        xb[p][quad][r][0] = mc;
#pragma unroll
        for (int k = 0; k < 4; ++k) xb[p][quad][r][1 + k] = mG[k];
    };
    // my row <- my row o (map of quad src): c += G c', G <- G G'
    auto compose_with = [&](int p, int src) {
        const T g0 = mG[0], g1 = mG[1], g2 = mG[2], g3 = mG[3];
        if (NW >= 4 || M > 1) {
            // throughput regime (many waves per SIMD): streamed by rows of the other map (rank-1
            // updates), one row = five LDS reads at a time: 5 numbers live instead of 20 -> fewer
            // registers, more waves per SIMD
            T n0, n1, n2, n3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T row[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) row[j] = xb[p][src][k][j];
                const T g = (k == 0) ? g0 : (k == 1) ? g1 : (k == 2) ? g2 : g3;
                cmac(mc, g, row[0]);
                if (k == 0) { n0 = g * row[1]; n1 = g * row[2]; n2 = g * row[3]; n3 = g * row[4]; }
                else { cmac(n0, g, row[1]); cmac(n1, g, row[2]); cmac(n2, g, row[3]); cmac(n3, g, row[4]); }
                __builtin_amdgcn_sched_barrier(0);      // keep the rows apart: the scheduler would hoist all 20 reads
            }
            mG[0] = n0; mG[1] = n1; mG[2] = n2; mG[3] = n3;
        } else {
            // latency regime (few waves): all 20 LDS reads in flight at once
            T oc[4], oG[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                oc[k] = xb[p][src][k][0];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) oG[k][cc] = xb[p][src][k][1 + cc];
            }
            cmac(mc, g0, oc[0]); cmac(mc, g1, oc[1]); cmac(mc, g2, oc[2]); cmac(mc, g3, oc[3]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                T t = g0 * oG[0][cc];
                cmac(t, g1, oG[1][cc]); cmac(t, g2, oG[2][cc]); cmac(t, g3, oG[3][cc]);
                mG[cc] = t;
            }
        }
    };
    // last scan step: only the offset c is used afterwards, G is dead (4 instead of 20 LDS reads,
    // 4 instead of 20 complex MACs)
    auto compose_c_only = [&](int p, int src) {
        cmac(mc, mG[0], xb[p][src][0][0]); cmac(mc, mG[1], xb[p][src][1][0]);
        cmac(mc, mG[2], xb[p][src][2][0]); cmac(mc, mG[3], xb[p][src][3][0]);
    };
    int p = 0;

    // ----------------------------- forward ---------------------------------
    // row r of a block map: c_r = (W b)[r+1],  G_rk = -(W[r+1][0] a_k + W[r+1][k+1] d_k)
    auto fwd_row = [&](int j) {
        T t = Wr[j][0] * b[j][0];
#pragma unroll
        for (int c = 1; c < 5; ++c) cmac(t, Wr[j][c], b[j][c]);
        mc = t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T g = Zero<T>::v();
            cmsc(g, Wr[j][0], av[j][k]);
            cmsc(g, Wr[j][k + 1], dv[j][k]);
            mG[k] = g;
        }
    };
    fwd_row(0);
    T u[4];             // z[1..4] of the block before the chunk
    if constexpr (CH) {
        // lane l <- lane l -+ 4 (the same row of the neighbouring quad); 4-block lines are one DPP row of 16 lanes
        auto from_below = [&](T v) -> T { return seg == 4 ? row_shift4<true>(v) : lane_shift4<true>(v); };
        T zc = mc;      // this quad's z[r + 1], assuming the quads below are final
#pragma unroll 1
        for (int st = 1; st < seg; ++st) {
            const T up = from_below(zc);
            T t0 = mc, t1 = mG[1] * quad_bcast<1>(up);
            cmac(t0, mG[0], quad_bcast<0>(up)); cmac(t1, mG[3], quad_bcast<3>(up)); cmac(t0, mG[2], quad_bcast<2>(up));
            zc = (ch > 0) ? t0 + t1 : mc;
        }
        const T up = from_below(zc);
        u[0] = quad_bcast<0>(up); u[1] = quad_bcast<1>(up); u[2] = quad_bcast<2>(up); u[3] = quad_bcast<3>(up);
        if (ch == 0) { u[0] = Zero<T>::v(); u[1] = Zero<T>::v(); u[2] = Zero<T>::v(); u[3] = Zero<T>::v(); }
    } else {
#pragma unroll
    for (int j = 1; j < M; ++j) {       // chunk map = block j after blocks 0..j-1
        publish(p);
        sync();
        fwd_row(j);
        compose_with(p, quad);
        p ^= 1;
    }
#pragma unroll 1
    for (int st = 1; st < seg; st <<= 1) {
        publish(p);
        sync();
        if (ch >= st) { if (2 * st < seg) compose_with(p, quad - st); else compose_c_only(p, quad - st); }
        p ^= 1;
    }
    publish(p);
    sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = (ch > 0) ? xb[p][quad - 1][k][0] : Zero<T>::v();
    p ^= 1;
    }
    // z_i = W_i (b_i - A_i z_{i-1}): lane r evaluates rows r+1 and 0; the quad hands z[1..4] on
    T z0[M], zr[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T y[5];
        y[0] = b[j][0];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cmsc(y[0], u[k], av[j][k]);
            T t = b[j][k + 1];
            cmsc(t, u[k], dv[j][k]);
            y[k + 1] = t;
        }
        z0[j] = W0[j][0] * y[0];
        zr[j] = Wr[j][0] * y[0];
#pragma unroll
        for (int c = 1; c < 5; ++c) { cmac(z0[j], W0[j][c], y[c]); cmac(zr[j], Wr[j][c], y[c]); }
        if (j + 1 < M) { u[0] = quad_bcast<0>(zr[j]); u[1] = quad_bcast<1>(zr[j]); u[2] = quad_bcast<2>(zr[j]); u[3] = quad_bcast<3>(zr[j]); }
    }

    QPL_TS(4);
    // ----------------------------- backward --------------------------------
    // v_i = A_i^T x_i (components 1..4) = g_i + H_i v_{i+1}:
    //   g_k = a_k z_0 + d_k z_k,   H_kl = -(a_k W[0][l] + d_k W[k][l])      (row k = r+1 in lane r)
    auto bwd_row = [&](int j) {
        const double ar = (r == 0) ? av[j][0] : (r == 1) ? av[j][1] : (r == 2) ? av[j][2] : av[j][3];
        const double dr = (r == 0) ? dv[j][0] : (r == 1) ? dv[j][1] : (r == 2) ? dv[j][2] : dv[j][3];
        T g = z0[j] * ar;
        cmac(g, zr[j], dr);
        mc = g;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            T h = Zero<T>::v();
            cmsc(h, W0[j][l + 1], ar);
            cmsc(h, Wr[j][l + 1], dr);
            mG[l] = h;
        }
    };
    bwd_row(M - 1);
    T v[4];             // A^T x of the block after the chunk
    if constexpr (CH) {
        auto from_above = [&](T w) -> T { return seg == 4 ? row_shift4<false>(w) : lane_shift4<false>(w); };
        T vc = mc;      // this quad's (A^T x)[r + 1], assuming the quads above are final
        const bool inner = ch + 1 < seg;
#pragma unroll 1
        for (int st = 1; st < seg; ++st) {
            const T dn = from_above(vc);
            T t0 = mc, t1 = mG[1] * quad_bcast<1>(dn);
            cmac(t0, mG[0], quad_bcast<0>(dn)); cmac(t1, mG[3], quad_bcast<3>(dn)); cmac(t0, mG[2], quad_bcast<2>(dn));
            vc = inner ? t0 + t1 : mc;
        }
        const T dn = from_above(vc);
        v[0] = quad_bcast<0>(dn); v[1] = quad_bcast<1>(dn); v[2] = quad_bcast<2>(dn); v[3] = quad_bcast<3>(dn);
        if (!inner) { v[0] = Zero<T>::v(); v[1] = Zero<T>::v(); v[2] = Zero<T>::v(); v[3] = Zero<T>::v(); }
    } else {
#pragma unroll
    for (int j = M - 2; j >= 0; --j) {  // chunk map = block j after blocks j+1.. (descending)
        publish(p);
        sync();
        bwd_row(j);
        compose_with(p, quad);
        p ^= 1;
    }
#pragma unroll 1
    for (int st = 1; st < seg; st <<= 1) {
        publish(p);
        sync();
        if (ch + st < seg) { if (2 * st < seg) compose_with(p, quad + st); else compose_c_only(p, quad + st); }
        p ^= 1;
    }
    publish(p);
    sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (ch + 1 < seg) ? xb[p][quad + 1][k][0] : Zero<T>::v();
    }
    QPL_TS(5);
    // x_i = z_i - W_i v
    T* eo = (a.e + boff_);
#pragma unroll
    for (int j = M - 1; j >= 0; --j) {
        T x0 = z0[j], xr = zr[j];
#pragma unroll
        for (int l = 0; l < 4; ++l) { cmsc(x0, W0[j][l + 1], v[l]); cmsc(xr, Wr[j][l + 1], v[l]); }
        const int i = ch * M + j;
        if (live && inl[j]) {       // (inside the line the clamped load offsets ARE the store offsets)
            if (r == 0) eo[dS0[j]] = x0;
            if (!lastb[j]) eo[dS[j]] = xr;
        }
        if (j > 0) {    // v of this block for the one before: v_k = a_k x_0 + d_k x_k
            const double ar = (r == 0) ? av[j][0] : (r == 1) ? av[j][1] : (r == 2) ? av[j][2] : av[j][3];
            const double dr = (r == 0) ? dv[j][0] : (r == 1) ? dv[j][1] : (r == 2) ? dv[j][2] : dv[j][3];
            T vr = x0 * ar;
            cmac(vr, xr, dr);
            v[0] = quad_bcast<0>(vr); v[1] = quad_bcast<1>(vr); v[2] = quad_bcast<2>(vr); v[3] = quad_bcast<3>(vr);
        }
    }
    QPL_TS(6);
    };      // one

    if constexpr (!HL) {
        one((u32)wg * (u32)lpg + (u32)(quad >> lseg), (u32)((a.mode == 0) ? a.cntA * a.cntB : a.cnt), (u32)a.t, (u32)a.jQ0);
#ifdef EMG3D_LAB
        if ((a.tile & 512) && blockIdx.x == 0 && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long tend = (long long)__builtin_readcyclecounter();
            printf("[qpl<%d,%d> seg %d lines %lld] args %lld loads issued %lld loads in + coefficients %lld forward %lld backward %lld stores issued %lld stores done %lld (cycles after entry)\n",
                   NW, M, seg, (long long)(a.cntA * a.cntB), qts[1] - qts[0], qts[2] - qts[0], qts[3] - qts[0], qts[4] - qts[0], qts[5] - qts[0], qts[6] - qts[0], tend - qts[0]);
        }
#endif
    } else {
        // Lexicographic order on a level of short lines: ONE workgroup per system walks through the hyperplanes
        // jP + 2 jQ = t (the lines of one are independent, consecutive ones are not) in rounds of lpg lines, a workgroup
        // barrier between them -- instead of one launch per hyperplane (7 us each for 2 us of work).  a.t / a.jQ0 = first /
        // last hyperplane, a.cnt = 1: descending (the first sweep runs backward, core.py:552, 569).
        const int nPm = (int)a.rs.nP - 1, nQm = (int)a.rs.nQ - 1;
        const int tmin = (int)a.t, tmax = (int)a.jQ0;
        for (int th = tmin; th <= tmax; ++th) {
            const int tt = a.cnt ? tmax - (th - tmin) : th;
            const int lo = tt - nPm;
            int jq0 = lo <= 0 ? 1 : (lo + 1) / 2;
            if (jq0 < 1) jq0 = 1;
            int jq1 = (tt - 1) / 2;
            if (jq1 > nQm) jq1 = nQm;
            const int n = jq1 - jq0 + 1;
            for (int base = 0; base < n; base += lpg) {
                one((u32)(quad >> lseg), (u32)min(n - base, lpg), (u32)tt, (u32)(jq0 + base));
                __threadfence_block();
                __syncthreads();
            }
        }
    }
#undef QPL_TS
}
