// Bandwidth-bound stencil kernels: residual (core.amat_x), restriction
// (core.restrict + model restriction), prolongation (solver.prolongation),
// norms.  One thread per output entry, x (the fastest storage axis) across
// lanes so that every global access of a wave is a contiguous run.
#pragma once
#include "common.hpp"
// (the few kernels below that are not templates are `static`: this header is included by several translation units)

#define EMG_BLOCK 256

// ---------------------------------------------------------------------------
// Residual  r = s - A e  (reference emg3d/core.py:29-177 + solver.py:980-1039)
//   MODE 0: r -= A e in place over the cell loops only (plain core.amat_x)
//   MODE 1: r  = s - A e stored, entries never touched by amat_x copy s
//   MODE 2: as 1 without storing r (norm only)
// When `partials` != nullptr each block writes sum |r|^2 of its entries.
// Thread (ix,iy) from a linear index inside one z-plane of the node index
// space (nNx x nNy), blockIdx.y = iz.
// ---------------------------------------------------------------------------
template <class T>
struct ResidualArgs {
    i64 nC[3];
    FieldLayout fl;
    T* r;
    const T* s;
    const T* e;
    const T* eta[3];
    const double* zeta;
    const double* h[3];
    const double* ih[3];      // 1 / h (the 42 divisions per cell of core.amat_x as multiplications)
    double* partials;
    Batch bt;                 // batched systems: r, s, e are [system][nE]; partials [system][blocks]
    int xcd = 0;              // block map (res_block_map): 0 plain, 1 z-slabs per XCD, 2 y-strips per XCD
    unsigned nbx = 0;         // blocks per plane
    int xs = 0;               // 1: e and s are the x-split working copies of the line sweeps (x index -> psplit, per
                              // component over its own x extent); r is always written in the reference layout
};
// x index of a field array with extent n along x: plain or parity-split
HD i64 xmap(int xs, i64 v, i64 n) { return xs ? psplit(v, n) : v; }

// Block maps of the residual kernels.  Workgroups are dealt round-robin to the 8 XCDs in flat order (x fastest), so the
// plain map (xcd = 0) spreads neighbouring blocks over all eight L2s.
//   xcd = 1 (slabs):  the w-th workgroup of a system takes slot (w mod 8) * (blocks / 8) + w / 8 -- every XCD walks
//                     through its own eighth of the z-planes;
//   xcd = 2 (strips): gridDim.x = 8 * ceil(nbx / 8); XCD c = blockIdx.x mod 8 owns the blocks [c nbx / 8, (c+1) nbx / 8)
//                     of EVERY plane (a strip of y), blockIdx.x / 8 counts inside the strip; surplus workgroups exit.
// nbx = blocks per plane (the layout of the partial sums).  Returns false for a surplus workgroup.
template <class A>
__device__ __forceinline__ bool res_block_map(const A& a, unsigned& bx, unsigned& bz) {
    bx = blockIdx.x; bz = blockIdx.y;
    if (a.xcd == 1) {
        const unsigned nbx = gridDim.x, tot = nbx * gridDim.y, w = blockIdx.y * nbx + blockIdx.x;
        const unsigned c = w & 7u, per = tot >> 3, rem = tot & 7u;
        const unsigned slot = c * per + (c < rem ? c : rem) + (w >> 3);
        bz = slot / nbx;
        bx = slot - bz * nbx;
    } else if (a.xcd == 2) {
        const unsigned c = blockIdx.x & 7u, b0 = (c * a.nbx) >> 3, b1 = ((c + 1) * a.nbx) >> 3;
        bx = b0 + (blockIdx.x >> 3);
        if (bx >= b1) return false;
    }
    return true;
}

template <class T, int MODE>
__global__ __launch_bounds__(EMG_BLOCK) void k_residual(ResidualArgs<T> a) {
    EMG_ARGS_BURST(EMG_S(a.nC[0]), EMG_S(a.nC[1]), EMG_S(a.nC[2]), EMG_S(a.r), EMG_S(a.s), EMG_S(a.e), EMG_S(a.eta[0]), EMG_S(a.eta[1]),
                   EMG_S(a.eta[2]), EMG_S(a.zeta), EMG_S(a.h[0]), EMG_S(a.h[1]), EMG_S(a.h[2]), EMG_S(a.ih[0]), EMG_S(a.ih[1]), EMG_S(a.ih[2]),
                   EMG_S(a.partials), EMG_S(a.bt.st), EMG_S(a.bt.mask), EMG_S(a.xcd), EMG_S(a.nbx), EMG_S(a.xs), EMG_S(a.fl.off[1]),
                   EMG_S(a.fl.off[2]));
    EMG_BATCH(z, a.bt);
    T* const r_ = a.r + boff_;
    const T* const s_ = a.s + boff_;
    const i64 nx = a.nC[0], ny = a.nC[1], nz = a.nC[2];
    const i64 nNx = nx + 1, nNy = ny + 1;
    unsigned bx, bz;
    if (!res_block_map(a, bx, bz)) return;
    const i64 lin = (i64)bx * EMG_BLOCK + threadIdx.x;
    const i64 iz = bz;
    i64 ix, iy;
    unlin2(lin, nNx, ix, iy);
    double acc = 0.0;
    if (iy < nNy) {
        const FieldLayout& f = a.fl;
        const T* e = a.e + boff_;
#define EX(i, j, k) e[f.off[0] + xmap(a.xs, (i), nx) * f.st[0][0] + (j) * f.st[0][1] + (k) * f.st[0][2]]
#define EY(i, j, k) e[f.off[1] + xmap(a.xs, (i), nNx) * f.st[1][0] + (j) * f.st[1][1] + (k) * f.st[1][2]]
#define EZ(i, j, k) e[f.off[2] + xmap(a.xs, (i), nNx) * f.st[2][0] + (j) * f.st[2][1] + (k) * f.st[2][2]]
#define ZT(i, j, k) a.zeta[(i) + nx * ((j) + ny * (k))]
#define CI(i, j, k) ((i) + nx * ((j) + ny * (k)))
        const bool incell = (ix < nx) && (iy < ny) && (iz < nz);
        T ax = Zero<T>::v(), ay = Zero<T>::v(), az = Zero<T>::v();  // (A e) rows
        if (incell) {
            const i64 ixm = ix > 0 ? ix - 1 : 0, ixp = ix + 1;
            const i64 iym = iy > 0 ? iy - 1 : 0, iyp = iy + 1;
            const i64 izm = iz > 0 ? iz - 1 : 0, izp = iz + 1;
            // reciprocal widths: a double-precision division is ~25 instructions, the kernel has 42 per cell
            const double hx = a.ih[0][ix], hxm = a.ih[0][ixm];
            const double hy = a.ih[1][iy], hym = a.ih[1][iym];
            const double hz = a.ih[2][iz], hzm = a.ih[2][izm];
            const T ex000 = EX(ix, iy, iz), ey000 = EY(ix, iy, iz), ez000 = EZ(ix, iy, iz);

            T v1pp = (EZ(ix, iyp, iz) - ez000) * hy - (EY(ix, iy, izp) - ey000) * hz;
            T v1mp = (ez000 - EZ(ix, iym, iz)) * hym - (EY(ix, iym, izp) - EY(ix, iym, iz)) * hz;
            T v1pm = (EZ(ix, iyp, izm) - EZ(ix, iy, izm)) * hy - (ey000 - EY(ix, iy, izm)) * hzm;
            T v2pp = (EX(ix, iy, izp) - ex000) * hz - (EZ(ixp, iy, iz) - ez000) * hx;
            T v2mp = (EX(ixm, iy, izp) - EX(ixm, iy, iz)) * hz - (ez000 - EZ(ixm, iy, iz)) * hxm;
            T v2pm = (ex000 - EX(ix, iy, izm)) * hzm - (EZ(ixp, iy, izm) - EZ(ix, iy, izm)) * hx;
            T v3pp = (EY(ixp, iy, iz) - ey000) * hx - (EX(ix, iyp, iz) - ex000) * hy;
            T v3mp = (ey000 - EY(ixm, iy, iz)) * hxm - (EX(ixm, iyp, iz) - EX(ixm, iy, iz)) * hy;
            T v3pm = (EY(ixp, iym, iz) - EY(ix, iym, iz)) * hx - (ex000 - EX(ix, iym, iz)) * hym;

            const double z000 = ZT(ix, iy, iz);
            v1pp *= ZT(ixm, iy, iz) + z000;
            v1mp *= ZT(ixm, iym, iz) + ZT(ix, iym, iz);
            v1pm *= ZT(ixm, iy, izm) + ZT(ix, iy, izm);
            v2pp *= ZT(ix, iym, iz) + z000;
            v2mp *= ZT(ixm, iym, iz) + ZT(ixm, iy, iz);
            v2pm *= ZT(ix, iym, izm) + ZT(ix, iy, izm);
            v3pp *= ZT(ix, iy, izm) + z000;
            v3mp *= ZT(ixm, iy, izm) + ZT(ixm, iy, iz);
            v3pm *= ZT(ix, iym, izm) + ZT(ix, iym, iz);

            T rrx = v3pp * hy - v3pm * hym - v2pp * hz + v2pm * hzm;
            T rry = v1pp * hz - v1pm * hzm - v3pp * hx + v3mp * hxm;
            T rrz = v2pp * hx - v2mp * hxm - v1pp * hy + v1mp * hym;

            const T stx = a.eta[0][CI(ix, iym, izm)] + a.eta[0][CI(ix, iym, iz)] +
                          a.eta[0][CI(ix, iy, izm)] + a.eta[0][CI(ix, iy, iz)];
            const T sty = a.eta[1][CI(ixm, iy, izm)] + a.eta[1][CI(ix, iy, izm)] +
                          a.eta[1][CI(ixm, iy, iz)] + a.eta[1][CI(ix, iy, iz)];
            const T stz = a.eta[2][CI(ixm, iym, iz)] + a.eta[2][CI(ix, iym, iz)] +
                          a.eta[2][CI(ixm, iy, iz)] + a.eta[2][CI(ix, iy, iz)];

            if (iy == 0 || iz == 0) rrx = Zero<T>::v();
            if (ix == 0 || iz == 0) rry = Zero<T>::v();
            if (ix == 0 || iy == 0) rrz = Zero<T>::v();

            ax = 0.5 * rrx - 0.25 * (stx * ex000);
            ay = 0.5 * rry - 0.25 * (sty * ey000);
            az = 0.5 * rrz - 0.25 * (stz * ez000);
        }
        const i64 px = f.off[0] + ix * f.st[0][0] + iy * f.st[0][1] + iz * f.st[0][2];
        const i64 py = f.off[1] + ix * f.st[1][0] + iy * f.st[1][1] + iz * f.st[1][2];
        const i64 pz = f.off[2] + ix * f.st[2][0] + iy * f.st[2][1] + iz * f.st[2][2];
        // the source's entries (the x-split copy when a.xs)
        const i64 qx = px + (xmap(a.xs, ix, nx) - ix) * f.st[0][0];
        const i64 qy = py + (xmap(a.xs, ix, nNx) - ix) * f.st[1][0];
        const i64 qz = pz + (xmap(a.xs, ix, nNx) - ix) * f.st[2][0];
        if (MODE == 0) {
            if (incell) {
                r_[px] -= ax; r_[py] -= ay; r_[pz] -= az;
            }
        } else {
            // entries exist for: fx ix<nx ; fy iy<ny ; fz iz<nz  (node-index space)
            if (ix < nx) {
                const T v = s_[qx] - ax;
                if (MODE == 1) r_[px] = v;
                acc += abs2(v);
            }
            if (iy < ny) {
                const T v = s_[qy] - ay;
                if (MODE == 1) r_[py] = v;
                acc += abs2(v);
            }
            if (iz < nz) {
                const T v = s_[qz] - az;
                if (MODE == 1) r_[pz] = v;
                acc += abs2(v);
            }
        }
#undef EX
#undef EY
#undef EZ
#undef ZT
#undef CI
    }
    if (MODE != 0 && a.partials != nullptr) {
        __shared__ double red[EMG_BLOCK / 64];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < EMG_BLOCK / 64; ++w) t += red[w];
            a.partials[((i64)b_ * gridDim.y + bz) * a.nbx + bx] = t;     // position of the SLOT: the sum order is fixed
        }
    }
}

// ---------------------------------------------------------------------------
// The same residual, KZ node planes per thread (MODE 1 / 2; the levels with millions of cells).  A thread walks up z
// and keeps what the next plane needs again in registers: of the 26 field, 8 zeta and 12 eta values a cell's rows read,
// 9 + 3 + 4 are the previous plane's (E_x, E_y at the upper nodes become the own ones, the own E_z, E_z(y+1), E_z(x+1)
// become the lower ones, ...), so a step issues 17 + 4 + 8 loads instead of 29 + 8 + 12.  Same statements in the same
// order as k_residual: identical results; the partial sums keep their per-plane layout (one slot per block and plane).
// gridDim.y = ceil((nz + 1) / KZ).
// ---------------------------------------------------------------------------
template <class T, int MODE, int KZ>
__global__ __launch_bounds__(EMG_BLOCK) void k_residual_zm(ResidualArgs<T> a) {
    static_assert(MODE == 1 || MODE == 2, "k_residual_zm: modes 1 and 2");
    EMG_ARGS_BURST(EMG_S(a.nC[0]), EMG_S(a.nC[1]), EMG_S(a.nC[2]), EMG_S(a.r), EMG_S(a.s), EMG_S(a.e), EMG_S(a.eta[0]), EMG_S(a.eta[1]),
                   EMG_S(a.eta[2]), EMG_S(a.zeta), EMG_S(a.h[0]), EMG_S(a.h[1]), EMG_S(a.h[2]), EMG_S(a.ih[0]), EMG_S(a.ih[1]), EMG_S(a.ih[2]),
                   EMG_S(a.partials), EMG_S(a.bt.st), EMG_S(a.bt.mask), EMG_S(a.xcd), EMG_S(a.nbx), EMG_S(a.xs), EMG_S(a.fl.off[1]),
                   EMG_S(a.fl.off[2]));
    EMG_BATCH(z, a.bt);
    T* __restrict__ const r_ = a.r + boff_;
    const T* __restrict__ const s_ = a.s + boff_;
    const T* __restrict__ const e = a.e + boff_;
    const i64 nx = a.nC[0], ny = a.nC[1], nz = a.nC[2];
    const i64 nNx = nx + 1, nNy = ny + 1, nNz = nz + 1;
    unsigned bx, bz;
    if (!res_block_map(a, bx, bz)) return;
    const i64 lin = (i64)bx * EMG_BLOCK + threadIdx.x;
    i64 ix, iy;
    unlin2(lin, nNx, ix, iy);
    const i64 iz0 = (i64)bz * KZ;
    const FieldLayout& f = a.fl;
    const bool inxy = iy < nNy;
    const bool cxy = inxy && ix < nx && iy < ny;
    double acc[KZ];
#pragma unroll
    for (int k = 0; k < KZ; ++k) acc[k] = 0.0;
    // x offsets of the field loads: X_ = the thread's own x, Xm_ / Xp_ its neighbours, per x extent (nx: E_x; nx + 1: E_y, E_z)
#define EX(X, j, k) e[f.off[0] + (X) * f.st[0][0] + (j) * f.st[0][1] + (k) * f.st[0][2]]
#define EY(X, j, k) e[f.off[1] + (X) * f.st[1][0] + (j) * f.st[1][1] + (k) * f.st[1][2]]
#define EZ(X, j, k) e[f.off[2] + (X) * f.st[2][0] + (j) * f.st[2][1] + (k) * f.st[2][2]]
#define ZT(i, j, k) a.zeta[(i) + nx * ((j) + ny * (k))]
#define CI(i, j, k) ((i) + nx * ((j) + ny * (k)))
    const i64 ixm = ix > 0 ? ix - 1 : 0, ixp = ix + 1;
    const i64 xc = xmap(a.xs, ix, nx), xcm = xmap(a.xs, ixm, nx);                              // E_x
    const i64 xn = xmap(a.xs, ix, nNx), xnm = xmap(a.xs, ixm, nNx), xnp = xmap(a.xs, ixp < nNx ? ixp : ix, nNx);   // E_y, E_z
    const i64 iym = iy > 0 ? iy - 1 : 0, iyp = iy + 1;
    double hx = 0, hxm = 0, hy = 0, hym = 0;
    // carried from plane to plane
    T ex000 = Zero<T>::v(), ex_xm = ex000, ex_zm = ex000;                  // EX(xc,iy,iz), EX(xcm,iy,iz), EX(xc,iy,izm)
    T ey000 = ex000, ey_ym = ex000, ey_zm = ex000;                         // EY(xn,iy,iz), EY(xn,iym,iz), EY(xn,iy,izm)
    T ez_zm = ex000, ez_yp_zm = ex000, ez_xp_zm = ex000;                   // EZ(xn,iy,izm), EZ(xn,iyp,izm), EZ(xnp,iy,izm)
    double zt_zm = 0, zt_xm_zm = 0, zt_ym_zm = 0;                          // ZT(ix,iy,izm), ZT(ixm,iy,izm), ZT(ix,iym,izm)
    T eta0_zm = ex000, eta0_ym_zm = ex000, eta1_zm = ex000, eta1_xm_zm = ex000;
    if (cxy && iz0 < nz) {
        const i64 izm = iz0 > 0 ? iz0 - 1 : 0;
        hx = a.ih[0][ix]; hxm = a.ih[0][ixm]; hy = a.ih[1][iy]; hym = a.ih[1][iym];
        ex000 = EX(xc, iy, iz0); ex_xm = EX(xcm, iy, iz0); ex_zm = EX(xc, iy, izm);
        ey000 = EY(xn, iy, iz0); ey_ym = EY(xn, iym, iz0); ey_zm = EY(xn, iy, izm);
        ez_zm = EZ(xn, iy, izm); ez_yp_zm = EZ(xn, iyp, izm); ez_xp_zm = EZ(xnp, iy, izm);
        zt_zm = ZT(ix, iy, izm); zt_xm_zm = ZT(ixm, iy, izm); zt_ym_zm = ZT(ix, iym, izm);
        eta0_zm = a.eta[0][CI(ix, iy, izm)]; eta0_ym_zm = a.eta[0][CI(ix, iym, izm)];
        eta1_zm = a.eta[1][CI(ix, iy, izm)]; eta1_xm_zm = a.eta[1][CI(ixm, iy, izm)];
    }
#pragma unroll
    for (int k = 0; k < KZ; ++k) {
        const i64 iz = iz0 + k;
        if (iz >= nNz) break;                                               // block-uniform
        T ax = Zero<T>::v(), ay = Zero<T>::v(), az = Zero<T>::v();
        if (cxy && iz < nz) {
            const i64 izm = iz > 0 ? iz - 1 : 0, izp = iz + 1;
            const double hz = a.ih[2][iz], hzm = a.ih[2][izm];
            const T ex_zp = EX(xc, iy, izp), ex_xm_zp = EX(xcm, iy, izp);
            const T ex_yp = EX(xc, iyp, iz), ex_xm_yp = EX(xcm, iyp, iz), ex_ym = EX(xc, iym, iz);
            const T ey_zp = EY(xn, iy, izp), ey_ym_zp = EY(xn, iym, izp);
            const T ey_xp = EY(xnp, iy, iz), ey_xm = EY(xnm, iy, iz), ey_xp_ym = EY(xnp, iym, iz);
            const T ez000 = EZ(xn, iy, iz), ez_yp = EZ(xn, iyp, iz), ez_ym = EZ(xn, iym, iz);
            const T ez_xp = EZ(xnp, iy, iz), ez_xm = EZ(xnm, iy, iz);
            const double z000 = ZT(ix, iy, iz), zt_xm = ZT(ixm, iy, iz), zt_ym = ZT(ix, iym, iz), zt_xm_ym = ZT(ixm, iym, iz);
            const T eta0_00 = a.eta[0][CI(ix, iy, iz)], eta0_ym = a.eta[0][CI(ix, iym, iz)];
            const T eta1_00 = a.eta[1][CI(ix, iy, iz)], eta1_xm = a.eta[1][CI(ixm, iy, iz)];

            T v1pp = (ez_yp - ez000) * hy - (ey_zp - ey000) * hz;
            T v1mp = (ez000 - ez_ym) * hym - (ey_ym_zp - ey_ym) * hz;
            T v1pm = (ez_yp_zm - ez_zm) * hy - (ey000 - ey_zm) * hzm;
            T v2pp = (ex_zp - ex000) * hz - (ez_xp - ez000) * hx;
            T v2mp = (ex_xm_zp - ex_xm) * hz - (ez000 - ez_xm) * hxm;
            T v2pm = (ex000 - ex_zm) * hzm - (ez_xp_zm - ez_zm) * hx;
            T v3pp = (ey_xp - ey000) * hx - (ex_yp - ex000) * hy;
            T v3mp = (ey000 - ey_xm) * hxm - (ex_xm_yp - ex_xm) * hy;
            T v3pm = (ey_xp_ym - ey_ym) * hx - (ex000 - ex_ym) * hym;

            v1pp *= zt_xm + z000;
            v1mp *= zt_xm_ym + zt_ym;
            v1pm *= zt_xm_zm + zt_zm;
            v2pp *= zt_ym + z000;
            v2mp *= zt_xm_ym + zt_xm;
            v2pm *= zt_ym_zm + zt_zm;
            v3pp *= zt_zm + z000;
            v3mp *= zt_xm_zm + zt_xm;
            v3pm *= zt_ym_zm + zt_ym;

            T rrx = v3pp * hy - v3pm * hym - v2pp * hz + v2pm * hzm;
            T rry = v1pp * hz - v1pm * hzm - v3pp * hx + v3mp * hxm;
            T rrz = v2pp * hx - v2mp * hxm - v1pp * hy + v1mp * hym;

            const T stx = eta0_ym_zm + eta0_ym + eta0_zm + eta0_00;
            const T sty = eta1_xm_zm + eta1_zm + eta1_xm + eta1_00;
            const T stz = a.eta[2][CI(ixm, iym, iz)] + a.eta[2][CI(ix, iym, iz)] +
                          a.eta[2][CI(ixm, iy, iz)] + a.eta[2][CI(ix, iy, iz)];

            if (iy == 0 || iz == 0) rrx = Zero<T>::v();
            if (ix == 0 || iz == 0) rry = Zero<T>::v();
            if (ix == 0 || iy == 0) rrz = Zero<T>::v();

            ax = 0.5 * rrx - 0.25 * (stx * ex000);
            ay = 0.5 * rry - 0.25 * (sty * ey000);
            az = 0.5 * rrz - 0.25 * (stz * ez000);

            // what the next plane reads again
            ex_zm = ex000; ex000 = ex_zp; ex_xm = ex_xm_zp;
            ey_zm = ey000; ey000 = ey_zp; ey_ym = ey_ym_zp;
            ez_zm = ez000; ez_yp_zm = ez_yp; ez_xp_zm = ez_xp;
            zt_zm = z000; zt_xm_zm = zt_xm; zt_ym_zm = zt_ym;
            eta0_zm = eta0_00; eta0_ym_zm = eta0_ym; eta1_zm = eta1_00; eta1_xm_zm = eta1_xm;
        }
        if (inxy) {
            const i64 px = f.off[0] + ix * f.st[0][0] + iy * f.st[0][1] + iz * f.st[0][2];
            const i64 py = f.off[1] + ix * f.st[1][0] + iy * f.st[1][1] + iz * f.st[1][2];
            const i64 pz = f.off[2] + ix * f.st[2][0] + iy * f.st[2][1] + iz * f.st[2][2];
            const i64 qx = px + (xc - ix) * f.st[0][0], qy = py + (xn - ix) * f.st[1][0], qz = pz + (xn - ix) * f.st[2][0];
            if (ix < nx) {
                const T v = s_[qx] - ax;
                if (MODE == 1) r_[px] = v;
                acc[k] += abs2(v);
            }
            if (iy < ny) {
                const T v = s_[qy] - ay;
                if (MODE == 1) r_[py] = v;
                acc[k] += abs2(v);
            }
            if (iz < nz) {
                const T v = s_[qz] - az;
                if (MODE == 1) r_[pz] = v;
                acc[k] += abs2(v);
            }
        }
    }
#undef EX
#undef EY
#undef EZ
#undef ZT
#undef CI
    if (a.partials != nullptr) {
        __shared__ double red[KZ][EMG_BLOCK / 64];
#pragma unroll
        for (int k = 0; k < KZ; ++k) {
            double t = acc[k];
            for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
            if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
        }
        __syncthreads();
        if (threadIdx.x < KZ && iz0 + threadIdx.x < nNz) {
            double t = 0.0;
            for (int w = 0; w < EMG_BLOCK / 64; ++w) t += red[threadIdx.x][w];
            a.partials[((i64)b_ * nNz + iz0 + threadIdx.x) * a.nbx + bx] = t;
        }
    }
}

// Deterministic final reduction: one block sums `n` partials in a fixed order
// and stores sqrt(sum) into out[slot].
static __global__ __launch_bounds__(EMG_BLOCK) void k_sum_sqrt(const double* partials, i64 n, double* out,
                                                        int slot) {
    // one block per system: out[slot * nb + b] = sqrt(sum of the n partials of system b)
    partials += (i64)blockIdx.x * n;
    __shared__ double red[EMG_BLOCK];
    double t = 0.0;
    // (eight loads in flight per thread, added in the SAME order as the plain loop: bit-identical sums; the loop was one dependent
    // memory round trip per 256 partials -- 100 us per 256^3 cycle for 66 k of them)
    i64 i = threadIdx.x;
    for (; i + 7 * (i64)EMG_BLOCK < n; i += 8 * (i64)EMG_BLOCK) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partials[i + (i64)k * EMG_BLOCK];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; i < n; i += EMG_BLOCK) t += partials[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = EMG_BLOCK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(i64)slot * gridDim.x + blockIdx.x] = sqrt(red[0]);
}

// sum |v|^2 partials of a flat buffer (for ||sfield||).
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_abs2_partials(const T* v, i64 n, double* partials) {
    double acc = 0.0;
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        acc += abs2(v[i]);
    __shared__ double red[EMG_BLOCK / 64];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < EMG_BLOCK / 64; ++w) t += red[w];
        partials[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// Restriction of the residual (reference emg3d/core.py:1586-1967), all seven
// sc_dir branches as per-axis flags; optional PEC on the coarse field
// (solver.py:898).  One launch per component c; thread = coarse edge.
// ---------------------------------------------------------------------------
template <class T>
struct RestrictArgs {
    i64 cnC[3], fnC[3];
    FieldLayout cfl, ffl;
    T* cr;
    T* ce;                  // coarse unknowns, zeroed alongside (nullptr: leave alone)
    const T* r;
    const double* w[3][3];  // [axis][l,0,r] on device (only for coarsened axes)
    int co[3];              // axis coarsened?
    int pec;
    Batch bt;               // batched systems: fine arrays
    i64 cbst = 0;           // ... coarse arrays: elements between systems
    int cxs = 0;            // 1: cr and ce are the coarse level's x-split working copies (see ResidualArgs::xs)
};

template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_restrict(RestrictArgs<T> a) {
    EMG_ARGS_BURST(EMG_S(a.cnC[0]), EMG_S(a.cnC[1]), EMG_S(a.cnC[2]), EMG_S(a.fnC[0]), EMG_S(a.fnC[1]), EMG_S(a.fnC[2]), EMG_S(a.cr), EMG_S(a.ce),
                   EMG_S(a.r), EMG_S(a.w[0][0]), EMG_S(a.w[0][1]), EMG_S(a.w[0][2]), EMG_S(a.w[1][0]), EMG_S(a.w[1][1]), EMG_S(a.w[1][2]),
                   EMG_S(a.w[2][0]), EMG_S(a.w[2][1]), EMG_S(a.w[2][2]), EMG_S(a.co[0]), EMG_S(a.co[1]), EMG_S(a.co[2]), EMG_S(a.pec),
                   EMG_S(a.bt.st), EMG_S(a.bt.mask), EMG_S(a.cbst), EMG_S(a.cxs));
    EMG_BATCH(z, a.bt);
    const T* const r_ = a.r + boff_;
    T* const cr_ = a.cr + (i64)b_ * a.cbst;
    T* const ce_ = a.ce ? a.ce + (i64)b_ * a.cbst : nullptr;
    const int c = blockIdx.y;          // component
    i64 cn[3];
    for (int q = 0; q < 3; ++q) cn[q] = (q == c) ? a.cnC[q] : a.cnC[q] + 1;
    const i64 lin = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x;
    if (lin >= cn[0] * cn[1] * cn[2]) return;
    i64 ci[3];
    unlin3(lin, cn[0], cn[1], cn[2], ci[0], ci[1], ci[2]);
    const int t1 = (c == 0) ? 1 : 0;
    const int t2 = (c == 2) ? 1 : 2;
    const i64 out = a.cfl.off[c] + xmap(a.cxs, ci[0], cn[0]) * a.cfl.st[c][0] + ci[1] * a.cfl.st[c][1] + ci[2] * a.cfl.st[c][2];
    if (ce_) ce_[out] = Zero<T>::v();      // zero initial guess of the coarse problem (solver.py:899)
    if (a.pec && (ci[t1] == 0 || ci[t1] == cn[t1] - 1 || ci[t2] == 0 || ci[t2] == cn[t2] - 1)) {
        cr_[out] = Zero<T>::v();
        return;
    }
    i64 f1[3], f2[3];
    double w1[3], w2[3];
    int n1 = 1, n2 = 1;
    if (a.co[t1]) {
        const i64 i = 2 * ci[t1], top = a.fnC[t1];  // fine nN-1 == fnC
        f1[0] = i; f1[1] = i > 0 ? i - 1 : 0; f1[2] = i + 1 < top ? i + 1 : top;
        w1[0] = a.w[t1][1][ci[t1]]; w1[1] = a.w[t1][0][ci[t1]]; w1[2] = a.w[t1][2][ci[t1]];
        n1 = 3;
    } else { f1[0] = ci[t1]; w1[0] = 1.0; }
    if (a.co[t2]) {
        const i64 i = 2 * ci[t2], top = a.fnC[t2];
        f2[0] = i; f2[1] = i > 0 ? i - 1 : 0; f2[2] = i + 1 < top ? i + 1 : top;
        w2[0] = a.w[t2][1][ci[t2]]; w2[1] = a.w[t2][0][ci[t2]]; w2[2] = a.w[t2][2][ci[t2]];
        n2 = 3;
    } else { f2[0] = ci[t2]; w2[0] = 1.0; }
    const i64 sc = a.ffl.st[c][c], s1 = a.ffl.st[c][t1], s2 = a.ffl.st[c][t2];
    const i64 basec = a.ffl.off[c] + (a.co[c] ? 2 * ci[c] : ci[c]) * sc;
    T acc = Zero<T>::v();
    for (int a1 = 0; a1 < n1; ++a1) {
        T inner = Zero<T>::v();
        for (int a2 = 0; a2 < n2; ++a2) {
            const i64 p = basec + f1[a1] * s1 + f2[a2] * s2;
            T v = r_[p];
            if (a.co[c]) v = v + r_[p + sc];
            if (n2 == 3) inner += w2[a2] * v; else inner = v;
        }
        if (n1 == 3) acc += w1[a1] * inner; else acc = inner;
    }
    cr_[out] = acc;
}

// ---------------------------------------------------------------------------
// Coarse model = sums of 8/4/2 fine cells (solver._restrict_model_parameters,
// emg3d/solver.py:1747-1784); summation order as in the reference.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_restrict_model(T* cp, const T* p, i64 cnx, i64 cny,
                                                              i64 cnz, i64 nx, i64 ny, int cox,
                                                              int coy, int coz) {
    const i64 lin = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x;
    if (lin >= cnx * cny * cnz) return;
    i64 cx, cy, cz;
    unlin3(lin, cnx, cny, cnz, cx, cy, cz);
    const i64 fx = cox ? 2 * cx : cx, fy = coy ? 2 * cy : cy, fz = coz ? 2 * cz : cz;
#define PF(dx, dy, dz) p[(fx + (dx)) + nx * ((fy + (dy)) + ny * (fz + (dz)))]
    T out;
    if (cox && coy && coz) {           // sc_dir 0
        out = PF(0, 0, 0) + PF(1, 0, 0);
        out += PF(0, 0, 1) + PF(1, 0, 1);
        out += PF(0, 1, 0) + PF(1, 1, 0);
        out += PF(0, 1, 1) + PF(1, 1, 1);
    } else if (!cox && coy && coz) {   // sc_dir 1
        out = PF(0, 0, 0) + PF(0, 1, 0);
        out += PF(0, 0, 1) + PF(0, 1, 1);
    } else if (cox && !coy && coz) {   // sc_dir 2
        out = PF(0, 0, 0) + PF(1, 0, 0);
        out += PF(0, 0, 1) + PF(1, 0, 1);
    } else if (cox && coy && !coz) {   // sc_dir 3
        out = PF(0, 0, 0) + PF(1, 0, 0);
        out += PF(0, 1, 0) + PF(1, 1, 0);
    } else if (cox) {                  // sc_dir 4
        out = PF(0, 0, 0) + PF(1, 0, 0);
    } else if (coy) {                  // sc_dir 5
        out = PF(0, 0, 0) + PF(0, 1, 0);
    } else {                           // sc_dir 6
        out = PF(0, 0, 0) + PF(0, 0, 1);
    }
#undef PF
    cp[lin] = out;
}

// ---------------------------------------------------------------------------
// Prolongation  e_f += P e_c  then PEC (solver.prolongation, emg3d/solver.py:
// 904-977; weights/corner order of RegularGridProlongator, 1409-1458).
// idx[a][j], wt[a][j]: per fine node j of axis a the lower coarse node and the
// normalised distance to it (host-computed with searchsorted semantics).
// One launch per component; thread = fine edge.
// ---------------------------------------------------------------------------
template <class T>
struct ProlongArgs {
    i64 fnC[3], cnC[3];
    FieldLayout ffl, cfl;
    T* e;
    const T* ce;
    const int* idx[3];
    const double* wt[3];
    int co[3];
    Batch bt;               // batched systems: fine arrays
    i64 cbst = 0;           // ... coarse arrays
    int fxs = 0;            // 1: the fine field is the x-split working copy (see ResidualArgs::xs)
    int cxs = 0;            // 1: so is the coarse field
};

template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_prolong(ProlongArgs<T> a) {
    EMG_ARGS_BURST(EMG_S(a.fnC[0]), EMG_S(a.fnC[1]), EMG_S(a.fnC[2]), EMG_S(a.cnC[0]), EMG_S(a.cnC[1]), EMG_S(a.cnC[2]), EMG_S(a.e), EMG_S(a.ce),
                   EMG_S(a.idx[0]), EMG_S(a.idx[1]), EMG_S(a.idx[2]), EMG_S(a.wt[0]), EMG_S(a.wt[1]), EMG_S(a.wt[2]), EMG_S(a.co[0]),
                   EMG_S(a.co[1]), EMG_S(a.co[2]), EMG_S(a.bt.st), EMG_S(a.bt.mask), EMG_S(a.cbst), EMG_S(a.fxs), EMG_S(a.cxs));
    EMG_BATCH(z, a.bt);
    T* const e_ = a.e + boff_;
    const T* const ce_ = a.ce + (i64)b_ * a.cbst;
    const int c = blockIdx.y;          // component
    i64 fn[3];
    for (int q = 0; q < 3; ++q) fn[q] = (q == c) ? a.fnC[q] : a.fnC[q] + 1;
    const i64 lin = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x;
    if (lin >= fn[0] * fn[1] * fn[2]) return;
    i64 fi[3];
    unlin3(lin, fn[0], fn[1], fn[2], fi[0], fi[1], fi[2]);
    const int t1 = (c == 0) ? 1 : 0;
    const int t2 = (c == 2) ? 1 : 2;
    const i64 p = a.ffl.off[c] + (a.fxs ? psplit(fi[0], fn[0]) : fi[0]) * a.ffl.st[c][0] + fi[1] * a.ffl.st[c][1] + fi[2] * a.ffl.st[c][2];
    if (fi[t1] == 0 || fi[t1] == fn[t1] - 1 || fi[t2] == 0 || fi[t2] == fn[t2] - 1) {
        e_[p] = Zero<T>::v();   // ensure_pec, fields.py:341-360
        return;
    }
    const i64 cc = a.co[c] ? fi[c] / 2 : fi[c];
    const i64 i1 = a.idx[t1][fi[t1]], i2 = a.idx[t2][fi[t2]];
    const double y1 = a.wt[t1][fi[t1]], y2 = a.wt[t2][fi[t2]];
    const i64 s1 = a.cfl.st[c][t1], s2 = a.cfl.st[c][t2];
    // offsets of the two coarse nodes along t1 and of the coarse cell along c (x-split coarse copy: x through psplit;
    // x is the field direction of component 0 and t1 of the others)
    const i64 cnx = (c == 0) ? a.cnC[0] : a.cnC[0] + 1;
    const i64 oc = ((a.cxs && c == 0) ? psplit(cc, cnx) : cc) * a.cfl.st[c][c];
    const i64 o1a = ((a.cxs && c != 0) ? psplit(i1, cnx) : i1) * s1;
    const i64 o1b = ((a.cxs && c != 0) ? psplit(i1 + 1 < cnx ? i1 + 1 : i1, cnx) : i1 + 1) * s1;
    const i64 q = a.cfl.off[c] + oc + i2 * s2;
    T hh = Zero<T>::v();
    hh += ce_[q + o1a] * ((1.0 * (1 - y1)) * (1 - y2));
    hh += ce_[q + o1a + s2] * ((1.0 * (1 - y1)) * y2);
    hh += ce_[q + o1b] * ((1.0 * y1) * (1 - y2));
    hh += ce_[q + o1b + s2] * ((1.0 * y1) * y2);
    e_[p] += hh;
}

// y = -y  (amatvec sign, solver.py:660) / generic scale.
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_negate(T* v, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        v[i] = -v[i];
}

// ---------------------------------------------------------------------------
// Vector primitives of the device-resident Krylov iteration (flat nE buffers):
// y += alpha x, y *= alpha, <a, b> = sum conj(a_i) b_i (complex) / sum a_i b_i (real).
// The reductions are deterministic: fixed grid, per-block partials, one final block.
// ---------------------------------------------------------------------------
HD c128 conj_mul(c128 a, c128 b) { return mk(a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re); }
HD double conj_mul(double a, double b) { return a * b; }
HD double imag_of(double) { return 0.0; }
HD double imag_of(c128 a) { return a.im; }

template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_axpy(T* __restrict__ y, const T* __restrict__ x, T alpha, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        y[i] = y[i] + alpha * x[i];
}
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_scale(T* __restrict__ y, T alpha, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        y[i] = alpha * y[i];
}
// partials[2 b], partials[2 b + 1] = real / imaginary part of block b's share of <a, b>
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_dot_partials(const T* __restrict__ a, const T* __restrict__ b, i64 n,
                                                            double* partials) {
    double re = 0.0, im = 0.0;
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK) {
        const T t = conj_mul(a[i], b[i]);
        re += real_of(t); im += imag_of(t);
    }
    __shared__ double red[2][EMG_BLOCK / 64];
    for (int o = 32; o > 0; o >>= 1) { re += __shfl_down(re, o, 64); im += __shfl_down(im, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = re; red[1][threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr = 0.0, ti = 0.0;
        for (int w = 0; w < EMG_BLOCK / 64; ++w) { tr += red[0][w]; ti += red[1][w]; }
        partials[2 * blockIdx.x] = tr; partials[2 * blockIdx.x + 1] = ti;
    }
}
// out[0], out[1] = sum of the nb (re, im) partial pairs, fixed order
static __global__ __launch_bounds__(EMG_BLOCK) void k_sum_pairs(const double* partials, i64 nb, double* out) {
    __shared__ double red[2][EMG_BLOCK];
    double tr = 0.0, ti = 0.0;
    for (i64 i = threadIdx.x; i < nb; i += EMG_BLOCK) { tr += partials[2 * i]; ti += partials[2 * i + 1]; }
    red[0][threadIdx.x] = tr; red[1][threadIdx.x] = ti;
    __syncthreads();
    for (int o = EMG_BLOCK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0]; }
}

// ---------------------------------------------------------------------------
// Magnetic field from the electric field, H = -curl E / (s mu_0)  (reference emg3d/fields.py:819-911):
// face-centred curl of the edge field, optionally scaled by zeta averaged onto the dual grid over the
// dual-cell volume (mu_r given), then the NEGATED value divided by s mu_0.  The arithmetic follows what
// NumPy evaluates there: complex / real and complex / complex are "multiply by the reciprocal" (Smith's
// division with rat, scl formed once: fields.py:865-875 divide by h, :911 by smu0), real / real is a true
// division.  Output layout [hx | hy | hz], F-ordered (nNx,nCy,nCz), (nCx,nNy,nCz), (nCx,nCy,nNz).
// Thread (ix, iy) of the node plane iz = blockIdx.y writes up to three values.  HBM bound:
// 48 B read + 48 B written per cell (+ 8 B zeta).
// ---------------------------------------------------------------------------
template <class T>
struct HFieldArgs {
    i64 nC[3];
    FieldLayout fl;         // layout of e
    const T* e;
    const double* zeta;     // nullptr: mu_r not given (fields.py:878)
    const double* h[3];
    const double* ih[3];    // 1.0 / h
    int hi;                 // Smith branch: |Re smu0| >= |Im smu0|
    double rat, scl;        // complex: Smith's ratio and scale; real: scl = smu0
    T* out;
};

#define EX(i, j, k) e[f.off[0] + (i) * f.st[0][0] + (j) * f.st[0][1] + (k) * f.st[0][2]]
#define EY(i, j, k) e[f.off[1] + (i) * f.st[1][0] + (j) * f.st[1][1] + (k) * f.st[1][2]]
#define EZ(i, j, k) e[f.off[2] + (i) * f.st[2][0] + (j) * f.st[2][1] + (k) * f.st[2][2]]
#define ZT(i, j, k) a.zeta[(i) + nx * ((j) + ny * (k))]
HD c128 hf_over_h(c128 v, double, double ih) { return mk(v.re * ih, v.im * ih); }
HD double hf_over_h(double v, double h, double) { return v / h; }
template <class T> HD T hf_finish(const HFieldArgs<T>& a, T v);
template <> HD c128 hf_finish<c128>(const HFieldArgs<c128>& a, c128 v) {
#pragma clang fp contract(off)      // NumPy's loops do not fuse; keeps the result bit-identical to the reference's
    const double vr = -v.re, vi = -v.im;
    return a.hi ? mk((vr + vi * a.rat) * a.scl, (vi - vr * a.rat) * a.scl)
                : mk((vr * a.rat + vi) * a.scl, (vi * a.rat - vr) * a.scl);
}
template <> HD double hf_finish<double>(const HFieldArgs<double>& a, double v) { return -v / a.scl; }

template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_hfield(HFieldArgs<T> a) {
#pragma clang fp contract(off)
    const i64 nx = a.nC[0], ny = a.nC[1], nz = a.nC[2];
    const i64 nNx = nx + 1, nNy = ny + 1;
    const i64 lin = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x;
    const i64 iz = blockIdx.y;
    i64 ix, iy;
    unlin2(lin, nNx, ix, iy);
    if (iy >= nNy) return;
    const FieldLayout& f = a.fl;
    const T* e = a.e;
    const bool cx = ix < nx, cy = iy < ny, cz = iz < nz;
    // dual-grid widths and the clipped cell indices either side of the node (fields.py:884-903)
    const i64 ixm = ix > 0 ? ix - 1 : 0, ixp = cx ? ix : nx - 1;
    const i64 iym = iy > 0 ? iy - 1 : 0, iyp = cy ? iy : ny - 1;
    const i64 izm = iz > 0 ? iz - 1 : 0, izp = cz ? iz : nz - 1;
    const double hx = cx ? a.h[0][ix] : 0.0, hy = cy ? a.h[1][iy] : 0.0, hz = cz ? a.h[2][iz] : 0.0;
    const double ihx = cx ? a.ih[0][ix] : 0.0, ihy = cy ? a.ih[1][iy] : 0.0, ihz = cz ? a.ih[2][iz] : 0.0;
    const i64 oy = nNx * ny * nz, oz = oy + nx * nNy * nz;
    if (cy && cz) {         // H_x = dEz/dy - dEy/dz on the face (node ix, cell iy, cell iz)
        T v = hf_over_h(EZ(ix, iy + 1, iz) - EZ(ix, iy, iz), hy, ihy) - hf_over_h(EY(ix, iy, iz + 1) - EY(ix, iy, iz), hz, ihz);
        if (a.zeta) {
            const double dx = ((ix > 0 ? a.h[0][ix - 1] : 0.0) + hx) / 2.;
            v *= ((ZT(ixm, iy, iz) + ZT(ixp, iy, iz)) / 2.) / (dx * hy * hz);
        }
        a.out[ix + nNx * (iy + ny * iz)] = hf_finish(a, v);
    }
    if (cx && cz) {         // H_y = dEx/dz - dEz/dx
        T v = hf_over_h(EX(ix, iy, iz + 1) - EX(ix, iy, iz), hz, ihz) - hf_over_h(EZ(ix + 1, iy, iz) - EZ(ix, iy, iz), hx, ihx);
        if (a.zeta) {
            const double dy = ((iy > 0 ? a.h[1][iy - 1] : 0.0) + hy) / 2.;
            v *= ((ZT(ix, iym, iz) + ZT(ix, iyp, iz)) / 2.) / (hx * dy * hz);
        }
        a.out[oy + ix + nx * (iy + nNy * iz)] = hf_finish(a, v);
    }
    if (cx && cy) {         // H_z = dEy/dx - dEx/dy
        T v = hf_over_h(EY(ix + 1, iy, iz) - EY(ix, iy, iz), hx, ihx) - hf_over_h(EX(ix, iy + 1, iz) - EX(ix, iy, iz), hy, ihy);
        if (a.zeta) {
            const double dz = ((iz > 0 ? a.h[2][iz - 1] : 0.0) + hz) / 2.;
            v *= ((ZT(ix, iy, izm) + ZT(ix, iy, izp)) / 2.) / (hx * hy * dz);
        }
        a.out[oz + ix + nx * (iy + ny * iz)] = hf_finish(a, v);
    }
}
#undef EX
#undef EY
#undef EZ
#undef ZT

// out = alpha * v for a REAL input array (eta = s mu_0 * (sigma V), models.py:631-658; source
// s = s mu_0 * vector, fields.py:624): the frequency enters the device problem as one scalar.
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_scale_real_to(T* __restrict__ out, const double* __restrict__ v, T alpha, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        out[i] = alpha * v[i];
}

// Is zeta the cell volume as the reference forms it -- zeta[i,j,k] == (hx_i * hy_j) * hz_k, bit for bit (emg3d/meshes.py:140-147,
// models.py:653-658 without mu_r)?  flag[0] is set to 1 by any cell that differs.
static __global__ __launch_bounds__(EMG_BLOCK) void k_zeta_is_volume(const double* __restrict__ zeta, const double* __restrict__ hx,
                                                             const double* __restrict__ hy, const double* __restrict__ hz,
                                                             i64 nx, i64 ny, i64 nz, int* flag) {
    const i64 n = nx * ny * nz;
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK) {
        const i64 ix = i % nx, iy = (i / nx) % ny, iz = i / (nx * ny);
        const double xy = hx[ix] * hy[iy];
        const double v = xy * hz[iz];
        bad = bad || !(zeta[i] == v);
    }
    if (bad) *flag = 1;
}

// sigma = 1 / rho in place (Model.conductivity for the 'Resistivity' mapping, reference models.py: an IEEE division, the
// same bits as NumPy's)
static __global__ __launch_bounds__(EMG_BLOCK) void k_recip_inplace(double* v, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK) v[i] = 1.0 / v[i];
}

// eta = (s mu_0 V) sigma as VolumeModel rounds it (reference models.py:631-658: `(smu0 * vol) * sigma`), from the cell
// volumes and the conductivities kept in HBM.  Frequency domain: s mu_0 = i b is purely imaginary and a complex x real
// product rounds the parts separately, so eta = (0 * t, t) with t = (b V) sigma; Laplace domain: t with b = s mu_0.
__device__ __forceinline__ double eta_of(double t, double) { return t; }
__device__ __forceinline__ c128 eta_of(double t, c128) { return mk(0.0 * t, t); }
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_eta_vs(T* __restrict__ out, const double* __restrict__ vol,
                                                     const double* __restrict__ sigma, double b, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK)
        out[i] = eta_of((b * vol[i]) * sigma[i], T());
}
// With displacement currents: eta = s mu_0 V (sigma - s eps_0 eps_r) (emg3d/models.py:631-647), evaluated as NumPy rounds
// it.  Laplace domain (s real): (b V) (sigma - c eps_r) with b = s mu_0, c = s eps_0.  Frequency domain (s mu_0 = i b,
// s eps_0 = i c): the complex products have one exactly-zero part each, so eta = ((b V)(c eps_r), (b V) sigma).
__device__ __forceinline__ double eta_eps_of(double p, double sig, double t, double) { return p * (sig - t); }
__device__ __forceinline__ c128 eta_eps_of(double p, double sig, double t, c128) { return mk(p * t, p * sig); }
template <class T>
__global__ __launch_bounds__(EMG_BLOCK) void k_eta_vs_eps(T* __restrict__ out, const double* __restrict__ vol,
                                                         const double* __restrict__ sigma, const double* __restrict__ epsr,
                                                         double b, double c, i64 n) {
    for (i64 i = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * EMG_BLOCK) {
        // (each product rounded on its own, as NumPy's array expressions are: no contraction into fused multiply-adds)
        const double p = b * vol[i];
        double t = c * epsr[i];
        asm volatile("" : "+v"(t));
        out[i] = eta_eps_of(p, sigma[i], t, T());
    }
}

// Swap the two fastest axes of a (a0, a1, nz) array, one z-plane per
// blockIdx.z, 32x32 tiles through LDS (+1 padding: conflict-free column reads).
// Block (32, 8).  SPLIT = 1: the destination's fastest axis is parity-split
// (to the working layout); SPLIT = -1: the SOURCE's fastest axis is parity-split
// (back to the reference layout); 2: BOTH are (between the two working layouts of the line sweeps: y-lines/z-lines copy
// <-> x-lines copy); 0: plain transpose.
template <class U, int SPLIT>
__global__ __launch_bounds__(256) void k_transpose01(U* __restrict__ dst, const U* __restrict__ src, i64 a0, i64 a1,
                                                     int nz, Batch bt) {
    __shared__ U tile[32][33];
    int z = blockIdx.z;
    if (bt.st) {        // batched systems: blockIdx.z = system * nz + plane
        const int b = z / nz;
        z -= b * nz;
        if (bt.mask && !bt.mask[b]) return;
        dst += (i64)b * bt.st; src += (i64)b * bt.st;
    }
    const i64 plane = a0 * a1 * (i64)z;
    const i64 i0 = (i64)blockIdx.x * 32, j0 = (i64)blockIdx.y * 32;
    for (int jj = threadIdx.y; jj < 32; jj += 8) {
        const i64 i = i0 + threadIdx.x, j = j0 + jj;
        if (i < a0 && j < a1) tile[jj][threadIdx.x] = src[plane + ((SPLIT < 0 || SPLIT == 2) ? psplit(i, a0) : i) + a0 * j];
    }
    __syncthreads();
    for (int ii = threadIdx.y; ii < 32; ii += 8) {
        const i64 j = j0 + threadIdx.x, i = i0 + ii;
        if (i < a0 && j < a1) dst[plane + (SPLIT > 0 ? psplit(j, a1) : j) + a1 * i] = tile[threadIdx.x][ii];
    }
}

// The three components of a field in ONE launch (plain transposition): blockIdx.z runs over the
// z-planes of component 0, then 1, then 2; tiles outside a component's (a0, a1) extent exit.
struct FieldTransArgs {
    i64 off[3], a0[3], a1[3];
    int nz[3];
};
template <class U>
__global__ __launch_bounds__(256) void k_transpose01_field(U* __restrict__ dst, const U* __restrict__ src, FieldTransArgs t,
                                                           Batch bt) {
    __shared__ U tile[32][33];
    int z = blockIdx.z, c = 0;
    if (bt.st) {
        const int nzt = t.nz[0] + t.nz[1] + t.nz[2], b = z / nzt;
        z -= b * nzt;
        if (bt.mask && !bt.mask[b]) return;
        dst += (i64)b * bt.st; src += (i64)b * bt.st;
    }
    if (z >= t.nz[0]) { z -= t.nz[0]; c = 1; }
    if (c == 1 && z >= t.nz[1]) { z -= t.nz[1]; c = 2; }
    const i64 a0 = t.a0[c], a1 = t.a1[c];
    const i64 i0 = (i64)blockIdx.x * 32, j0 = (i64)blockIdx.y * 32;
    if (i0 >= a0 || j0 >= a1) return;
    const i64 plane = t.off[c] + a0 * a1 * (i64)z;
    for (int jj = threadIdx.y; jj < 32; jj += 8) {
        const i64 i = i0 + threadIdx.x, j = j0 + jj;
        if (i < a0 && j < a1) tile[jj][threadIdx.x] = src[plane + i + a0 * j];
    }
    __syncthreads();
    for (int ii = threadIdx.y; ii < 32; ii += 8) {
        const i64 j = j0 + threadIdx.x, i = i0 + ii;
        if (i < a0 && j < a1) dst[plane + j + a1 * i] = tile[threadIdx.x][ii];
    }
}

// Parity split (DIR = 1) / un-split (DIR = -1) of the fastest axis of an
// (n0, rows) array.  Thread per element, grid-stride.
template <class U, int DIR>
__global__ __launch_bounds__(EMG_BLOCK) void k_split0(U* __restrict__ dst, const U* __restrict__ src, i64 n0, i64 rows,
                                                      Batch bt) {
    EMG_BATCH(y, bt); dst += boff_; src += boff_;
    const i64 n = n0 * rows;
    for (i64 t = (i64)blockIdx.x * EMG_BLOCK + threadIdx.x; t < n; t += (i64)gridDim.x * EMG_BLOCK) {
        const i64 row = t / n0, i = t - row * n0;
        if (DIR > 0) dst[row * n0 + psplit(i, n0)] = src[t];
        else dst[t] = src[row * n0 + psplit(i, n0)];
    }
}
