// Adjoint-state gradient pieces on the device:
//   k_edges2cell  = maps.edges2cellaverages (reference emg3d/maps.py:578-630): edge values -> volume-weighted cell
//                   averages, one output array per component;
//   k_gradient    = optimize.gradient for one (source, frequency) pair on its computational grid (reference
//                   emg3d/optimize.py:176-199): -Re(lambda E s mu_0) on the edges, mapped to cells, components added.
// Gather form: one thread per CELL sums the contributions of its 12 edges in the order in which the reference's
// loops (iz, iy, ix ascending, four statements per edge) add them -- deterministic and equal to the reference's
// accumulation order, no atomics.
#pragma once
#include "common.hpp"

// What the reference adds into cell (j0, j1, j2) for component c: sum over the cell's four c-edges (and the
// duplicated boundary statements) of vol * f / 4.  F(i0, i1, i2) returns the edge value.
template <class V, class F>
__device__ __forceinline__ V e2c_component(int c, const i64 j[3], const i64 nC[3], double vol, F f) {
    const int t1 = (c == 0) ? 1 : 0, t2 = (c == 2) ? 1 : 2;       // transverse axes, t1 < t2 (t2 is the outer loop)
    V acc = V();
    for (i64 e2 = j[t2]; e2 <= j[t2] + 1; ++e2) {
        const i64 m2 = e2 > 0 ? e2 - 1 : 0, p2 = e2 < nC[t2] - 1 ? e2 : nC[t2] - 1;
        for (i64 e1 = j[t1]; e1 <= j[t1] + 1; ++e1) {
            const i64 m1 = e1 > 0 ? e1 - 1 : 0, p1 = e1 < nC[t1] - 1 ? e1 : nC[t1] - 1;
            i64 e[3];
            e[c] = j[c]; e[t1] = e1; e[t2] = e2;
            const V v = (vol * f(e[0], e[1], e[2])) / 4.0;
            // the four statements of the reference, in its order: (m1, m2), (p1, m2), (m1, p2), (p1, p2)
            if (m1 == j[t1] && m2 == j[t2]) acc += v;
            if (p1 == j[t1] && m2 == j[t2]) acc += v;
            if (m1 == j[t1] && p2 == j[t2]) acc += v;
            if (p1 == j[t1] && p2 == j[t2]) acc += v;
        }
    }
    return acc;
}

template <class T>
struct E2CArgs {
    i64 nC[3];
    FieldLayout fl;
    const T* f;              // [fx|fy|fz]
    const double* vol;       // F-ordered (nx, ny, nz) cell volumes, or NULL: hx*hy*hz from h
    const double* h[3];
    T* out[3];               // out_x, out_y, out_z (+=)
};

template <class T>
__global__ void k_edges2cell(E2CArgs<T> a) {
    const i64 n = a.nC[0] * a.nC[1] * a.nC[2];
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const i64 j[3] = {idx % a.nC[0], (idx / a.nC[0]) % a.nC[1], idx / (a.nC[0] * a.nC[1])};
    const double vol = a.vol ? a.vol[idx] : (a.h[0][j[0]] * a.h[1][j[1]]) * a.h[2][j[2]];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T* comp = a.f + a.fl.off[c];
        const i64 s0 = a.fl.st[c][0], s1 = a.fl.st[c][1], s2 = a.fl.st[c][2];
        a.out[c][idx] += e2c_component<T>(c, j, a.nC, vol, [&](i64 i0, i64 i1, i64 i2) { return comp[i0 * s0 + i1 * s1 + i2 * s2]; });
    }
}

__device__ __forceinline__ double grad_prod(double b, double e, double s, double) { return -((b * e) * s); }
__device__ __forceinline__ double grad_prod(c128 b, c128 e, double sr, double si) {
    const c128 be = b * e;                         // -real(bfield * efield * smu0), optimize.py:181-184
    return -(be.re * sr - be.im * si);
}

// grad[cell] = sum_c edges2cellaverages_c( -Re(b * e * smu0) ), vol = (hx*hy)*hz (meshes cell_volumes)
template <class T>
__global__ void k_gradient(i64 n0, i64 n1, i64 n2, FieldLayout fl, const T* e, const T* b, double sr, double si,
                           const double* h0, const double* h1, const double* h2, double* grad) {
    const i64 nC[3] = {n0, n1, n2};
    const i64 n = n0 * n1 * n2;
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const i64 j[3] = {idx % n0, (idx / n0) % n1, idx / (n0 * n1)};
    const double vol = (h0[j[0]] * h1[j[1]]) * h2[j[2]];
    double g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T* ec = e + fl.off[c];
        const T* bc = b + fl.off[c];
        const i64 s0 = fl.st[c][0], s1 = fl.st[c][1], s2 = fl.st[c][2];
        g[c] = e2c_component<double>(c, j, nC, vol, [&](i64 i0, i64 i1, i64 i2) {
            const i64 o = i0 * s0 + i1 * s1 + i2 * s2;
            return grad_prod(bc[o], ec[o], sr, si);
        });
    }
    grad[idx] = (g[0] + g[1]) + g[2];          // grad_x + grad_y + grad_z, optimize.py:199
}
