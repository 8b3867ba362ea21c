// Receiver extraction on the device: fields.get_receiver_response (reference emg3d/fields.py:733-817) and
// maps.interp3d (reference emg3d/maps.py:179-276; mode='constant').
//
// The cubic branch of interp3d is SciPy in the reference (maps.py:257-272): interp1d(kind='cubic') -- the
// not-a-knot cubic spline through (points[i], i), extrapolated -- turns coordinates into fractional indices
// (O(n) host work, notaknot_index_coords below), and ndimage.map_coordinates(order=3, mode='constant', cval)
// interpolates: a separable cubic B-spline prefilter over the WHOLE array (pole sqrt(3) - 2, mirror boundary
// initialisation; k_spline_filter_axis: one recursion per line, three passes) followed by a 4 x 4 x 4 weighted
// sum per point (k_spline_eval); points outside [0, n-1] get cval.  The linear branch is
// RegularGridInterpolator(method='linear', bounds_error=False, fill_value) (k_linear_eval).
//
// With the field resident in HBM only the responses (16 bytes per receiver) cross PCIe / xGMI instead of the
// field (102 MB at 128^3): SURVEY 8f rank 3.
#pragma once
#include <cmath>
#include <vector>
#include "common.hpp"

#define EMG_RCV_BLOCK 128

// ---- host: fractional index of xi in the grid x (not-a-knot cubic spline through (x_i, i)) ------------------
inline void notaknot_index_coords(const double* x, i64 n, const double* xi, i64 m, double* out) {
    std::vector<double> h(n - 1), M(n, 0.0);
    for (i64 i = 0; i + 1 < n; ++i) h[i] = x[i + 1] - x[i];
    // dense (n x n) system of the second derivatives with partial pivoting: n is a grid dimension (<= ~1000),
    // solved once per axis and field component
    std::vector<double> A((size_t)n * n, 0.0), r(n, 0.0);
    auto a = [&](i64 i, i64 j) -> double& { return A[(size_t)i * n + j]; };
    for (i64 i = 1; i + 1 < n; ++i) {
        a(i, i - 1) = h[i - 1]; a(i, i) = 2 * (h[i - 1] + h[i]); a(i, i + 1) = h[i];
        r[i] = 6 * (1.0 / h[i] - 1.0 / h[i - 1]);          // y_i = i: (y_{i+1} - y_i) = 1
    }
    a(0, 0) = h[1]; a(0, 1) = -(h[0] + h[1]); a(0, 2) = h[0];
    a(n - 1, n - 3) = h[n - 2]; a(n - 1, n - 2) = -(h[n - 3] + h[n - 2]); a(n - 1, n - 1) = h[n - 3];
    // banded elimination with row pivoting inside a window of 3 rows (the matrix is tridiagonal plus the
    // two 3-entry end rows)
    for (i64 c = 0; c < n; ++c) {
        i64 p = c;
        const i64 rmax = std::min<i64>(n - 1, c + 2);
        for (i64 q = c + 1; q <= rmax; ++q) if (std::fabs(a(q, c)) > std::fabs(a(p, c))) p = q;
        if (p != c) {
            const i64 jmax = std::min<i64>(n - 1, c + 4);
            for (i64 j = c; j <= jmax; ++j) std::swap(a(c, j), a(p, j));
            std::swap(r[c], r[p]);
        }
        for (i64 q = c + 1; q <= rmax; ++q) {
            const double f = a(q, c) / a(c, c);
            if (f == 0.0) continue;
            const i64 jmax = std::min<i64>(n - 1, c + 4);
            for (i64 j = c; j <= jmax; ++j) a(q, j) -= f * a(c, j);
            r[q] -= f * r[c];
        }
    }
    for (i64 c = n - 1; c >= 0; --c) {
        double s = r[c];
        const i64 jmax = std::min<i64>(n - 1, c + 4);
        for (i64 j = c + 1; j <= jmax; ++j) s -= a(c, j) * M[j];
        M[c] = s / a(c, c);
    }
    for (i64 k = 0; k < m; ++k) {
        const double v = xi[k];
        i64 i = (i64)(std::lower_bound(x, x + n, v) - x) - 1;     // searchsorted(x, v) - 1
        if (i < 0) i = 0;
        if (i > n - 2) i = n - 2;
        const double t0 = v - x[i], t1 = x[i + 1] - v, hi = h[i];
        out[k] = (M[i] * t1 * t1 * t1 + M[i + 1] * t0 * t0 * t0) / (6 * hi) + ((double)i / hi - M[i] * hi / 6) * t1 +
                 ((double)(i + 1) / hi - M[i + 1] * hi / 6) * t0;
        // at the end points the interpolating spline IS its data value (SciPy's B-spline form returns exactly 0 and
        // n - 1 there); the cancelling terms above may leave +-1e-17, which would push a receiver on the first / last
        // point of the trimmed grid outside (cval = NaN)
        if (v == x[0]) out[k] = 0.0;
        if (v == x[n - 1]) out[k] = (double)(n - 1);
    }
}

// ---- device ---------------------------------------------------------------------------------------------------
// dst (n0-2, n1-2, n2-2) <- src[1:-1, 1:-1, 1:-1] of an F-ordered (n0, n1, n2) array (fields.py:799, 812)
template <class T>
__global__ void k_trim_copy(T* dst, const T* src, i64 n0, i64 n1, i64 n2) {
    const i64 m0 = n0 - 2, m1 = n1 - 2, m2 = n2 - 2;
    const i64 tot = m0 * m1 * m2;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (i64)gridDim.x * blockDim.x) {
        const i64 i0 = idx % m0, i1 = (idx / m0) % m1, i2 = idx / (m0 * m1);
        dst[idx] = src[(i0 + 1) + n0 * ((i1 + 1) + n1 * (i2 + 1))];
    }
}

// In-place cubic B-spline prefilter along `axis` of an F-ordered (n0, n1, n2) array: thread per line.
// scipy.ndimage spline_filter1d, order 3, mirror initialisation (what map_coordinates(mode='constant') uses):
// gain, causal initialisation, causal recursion, anti-causal initialisation, anti-causal recursion.
// reflect != 0: the half-sample-symmetric ("reflect") initialisation that scipy uses for mode='nearest' / 'reflect'.
template <class T>
__global__ void k_spline_filter_axis(T* c, i64 n0, i64 n1, i64 n2, int axis, int reflect) {
    const i64 n = axis == 0 ? n0 : axis == 1 ? n1 : n2;
    const i64 nlines = (n0 * n1 * n2) / n;
    const i64 line = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nlines || n < 2) return;
    i64 base, st;
    if (axis == 0) { base = line * n0; st = 1; }
    else if (axis == 1) { const i64 i0 = line % n0, i2 = line / n0; base = i0 + n0 * n1 * i2; st = n0; }
    else { base = line; st = n0 * n1; }
    const double z = sqrt(3.0) - 2.0;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    T* p = c + base;
    for (i64 i = 0; i < n; ++i) p[i * st] *= gain;
    if (reflect) {
        const double zn = pow(z, (double)n);
        const T first = p[0];
        T c0 = p[0] + zn * p[(n - 1) * st];
        double zi = z;
        for (i64 i = 1; i < n; ++i) {
            c0 += zi * (p[i * st] + zn * p[(n - 1 - i) * st]);
            zi *= z;
        }
        c0 *= z / (1.0 - zn * zn);
        p[0] = c0 + first;
    } else {
        const double zn1 = pow(z, (double)(n - 1));
        T c0 = p[0] + zn1 * p[(n - 1) * st];
        double zi = z;
        for (i64 i = 1; i < n - 1; ++i) {
            c0 += zi * (p[i * st] + zn1 * p[(n - 1 - i) * st]);
            zi *= z;
        }
        p[0] = c0 / (1.0 - zn1 * zn1);
    }
    T prev = p[0];
    for (i64 i = 1; i < n; ++i) {
        T v = p[i * st];
        v += z * prev;
        p[i * st] = v;
        prev = v;
    }
    const T last = reflect ? p[(n - 1) * st] * (z / (z - 1.0))
                           : (z * p[(n - 2) * st] + p[(n - 1) * st]) * (z / (z * z - 1.0));
    p[(n - 1) * st] = last;
    T nxt = last;
    for (i64 i = n - 2; i >= 0; --i) {
        const T v = z * (nxt - p[i * st]);
        p[i * st] = v;
        nxt = v;
    }
}

__device__ __forceinline__ void bspline3_weights(double t, double w[4]) {
    w[1] = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
    const double u = 1.0 - t;
    w[2] = (u * u * (u - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = u * u * u / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}
__device__ __forceinline__ i64 mirror_index(i64 j, i64 n) {
    if (n == 1) return 0;
    const i64 s2 = 2 * n - 2;
    if (j < 0) {
        j = s2 * (i64)(-j / s2) + j;
        j = (j <= 1 - n) ? j + s2 : -j;
    } else if (j >= n) {
        j -= s2 * (i64)(j / s2);
        if (j >= n) j = s2 - j;
    }
    return j;
}
// Index into the half-sample symmetric extension (d c b a | a b c d | d c b a; scipy's NI_EXTEND_REFLECT), for ANY j:
// scipy maps the coordinate into the array first and only ever reflects indices next to it; here the stencil stays at
// the caller's coordinate, so the rule has to hold far away too (period 2 n).
__device__ __forceinline__ i64 reflect_index(i64 j, i64 n) {
    const i64 s2 = 2 * n;
    i64 m = j % s2;
    if (m < 0) m += s2;
    return m < n ? m : s2 - 1 - m;
}
template <class T> __device__ __forceinline__ T fill_of(double v);
template <> __device__ __forceinline__ double fill_of<double>(double v) { return v; }
template <> __device__ __forceinline__ c128 fill_of<c128>(double v) { return mk(v, v); }

// out[r] += fac[r] * (cubic B-spline interpolant of `coef` at the fractional indices coords[a][r]); thread per point.
// fac == nullptr: out[r] = value.
template <class T>
// edge = 0 (mode='constant'): points outside [0, n-1] get cval, stencil indices beyond the array are mirrored;
// edge = 1 (mode='nearest', on the pre-padded array) / 2 (mode='mirror', and 'wrap' on coordinates wrapped by the caller) /
// 3 (mode='reflect'): no point is outside -- the stencil sits at the coordinate and indices beyond the array take the edge
// coefficient / are mirrored / are reflected (scipy's NI_EXTEND_NEAREST / _MIRROR / _REFLECT).
__global__ __launch_bounds__(EMG_RCV_BLOCK) void k_spline_eval(T* out, const T* coef, i64 n0, i64 n1, i64 n2, const double* coords, const double* fac,
                              i64 npts, double cval, int edge) {
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npts) return;
    const i64 nn[3] = {n0, n1, n2};
    i64 idx[3][4];
    double w[3][4];
    bool outside = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double cc = coords[a * npts + r];
        if (edge) {
            if (!(cc == cc) || fabs(cc) > 1e15) outside = true;             // NaN (or beyond integer range)
            if (edge == 1) cc = cc < -4.0 ? -4.0 : (cc > (double)(nn[a] + 3) ? (double)(nn[a] + 3) : cc);   // far away: all four indices clamp alike
        } else if (!(cc >= 0.0 && cc <= (double)(nn[a] - 1))) outside = true;                // NaN coordinates count as outside
        const double fl = floor(cc);
        bspline3_weights(cc - fl, w[a]);
        const i64 start = (i64)fl - 1;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const i64 j = start + l;
            idx[a][l] = outside ? 0 : edge == 1 ? (j < 0 ? 0 : (j > nn[a] - 1 ? nn[a] - 1 : j))
                                    : edge == 3 ? reflect_index(j, nn[a]) : mirror_index(j, nn[a]);
        }
    }
    T val;
    if (outside) {
        val = fill_of<T>(cval);
    } else {
        // the 64 coefficients are loaded together, then summed in the fixed order (written as one loop the float64
        // instantiation issued one load per term and waited for it: 56 dependent round trips)
        T c[64];
#pragma unroll
        for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
            for (int a1 = 0; a1 < 4; ++a1)
#pragma unroll
                for (int a2 = 0; a2 < 4; ++a2)
                    c[(a0 * 4 + a1) * 4 + a2] = coef[idx[0][a0] + n0 * (idx[1][a1] + n1 * idx[2][a2])];
        __builtin_amdgcn_sched_barrier(0);
        val = Zero<T>::v();
#pragma unroll
        for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
            for (int a1 = 0; a1 < 4; ++a1)
#pragma unroll
                for (int a2 = 0; a2 < 4; ++a2)
                    val += c[(a0 * 4 + a1) * 4 + a2] * (w[0][a0] * w[1][a1] * w[2][a2]);
    }
    if (fac) out[r] += fac[r] * val;
    else out[r] = val;
}

// Trilinear interpolation, thread per point: ii / tt = interval index and normalised distance per axis (host:
// searchsorted - 1 clipped, as RegularGridInterpolator), inside[r] = 0: the point gets `fill`.
// The array is addressed with explicit offset / strides, so the trimmed view [1:-1]^3 needs no copy.
template <class T>
__global__ void k_linear_eval(T* out, const T* values, i64 s0, i64 s1, i64 s2, const int* ii, const double* tt,
                              const int* inside, const double* fac, i64 npts, double fill) {
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npts) return;
    T val;
    if (!inside[r]) {
        val = Zero<T>::v();
        add_real(val, fill);            // fill_value is a real scalar (complex values: fill + 0j)
    } else {
        const i64 i0 = ii[r], i1 = ii[npts + r], i2 = ii[2 * npts + r];
        const double t0 = tt[r], t1 = tt[npts + r], t2 = tt[2 * npts + r];
        val = Zero<T>::v();
        for (int a0 = 0; a0 < 2; ++a0)
            for (int a1 = 0; a1 < 2; ++a1)
                for (int a2 = 0; a2 < 2; ++a2) {
                    const double w = (a0 ? t0 : 1 - t0) * (a1 ? t1 : 1 - t1) * (a2 ? t2 : 1 - t2);
                    val += values[(i0 + a0) * s0 + (i1 + a1) * s1 + (i2 + a2) * s2] * w;
                }
    }
    if (fac) out[r] += fac[r] * val;
    else out[r] = val;
}

// ---- host driver ----------------------------------------------------------------------------------------------
// One field component on the device: F-ordered (n[0], n[1], n[2]) array and the grid vector of every axis.
template <class T>
struct RcvComp {
    const T* dev = nullptr;
    i64 n[3] = {0, 0, 0};
    std::vector<double> pts[3];
};

// interp3d(points, values, xi, method, fill_value, 'constant', cval) of a device-resident array (maps.py:179-276):
// out_dev[r] (+)= fac[r] * value.  `off`/(s0,s1,s2): the view of `values` that `pts` describe (the receivers pass the
// [1:-1]^3 view: off = 1 + n0 + n0 n1).  scratch: >= prod(n) entries for the spline coefficients (cubic only).
template <class T>
int interp3d_device(hipStream_t st, const T* values, const i64 n[3], i64 off, i64 s0, i64 s1, i64 s2,
                    const std::vector<double> pts[3], i64 npts, const double* xi /* [3][npts] */, int method,
                    bool has_fill, double fill, double cval, const double* fac_host, T* scratch, T* out_dev) {
    for (int a = 0; a < 3; ++a) if ((i64)pts[a].size() != n[a] || n[a] < 1) return -2;
    // method: 0 linear, 1 cubic, 2 / 3 / 4 cubic with `xi` already in INDEX coordinates of `values` (the caller has applied the
    // not-a-knot index spline and a boundary mode of scipy.ndimage.map_coordinates: emg3d_amd/maps.py); 3: the arithmetic of
    // mode='nearest' on the pre-padded array ("reflect" prefilter initialisation, indices clamped, nothing is outside)
    for (int a = 0; a < 3; ++a) if (n[a] < 4 && method < 2) method = 0;             // maps.py:238-240
    if (method >= 2) for (int a = 0; a < 3; ++a) if (n[a] < 4) return -2;
    const unsigned blocks = (unsigned)((npts + EMG_RCV_BLOCK - 1) / EMG_RCV_BLOCK);
    double* dfac = nullptr;
    DevBlock tmpb;
    const size_t nb = (size_t)npts * (3 * sizeof(double) + sizeof(double) + 3 * sizeof(int) + sizeof(int)) + 256;
    HIP_TRY(tmpb.alloc(nb));
    char* tmp = tmpb.get<char>();
    double* dco = (double*)tmp;                         // coords / tt
    dfac = dco + 3 * npts;
    int* dii = (int*)(dfac + npts);
    int* dins = dii + 3 * npts;
    if (fac_host) HIP_TRY(hipMemcpyAsync(dfac, fac_host, (size_t)npts * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = 0;
    if (method >= 1) {
        std::vector<double> co((size_t)3 * npts);
        for (int a = 0; a < 3; ++a) {
            if (method >= 2) std::copy(xi + a * npts, xi + (a + 1) * npts, co.data() + a * npts);
            else notaknot_index_coords(pts[a].data(), n[a], xi + a * npts, npts, co.data() + a * npts);
        }
        HIP_TRY(hipMemcpyAsync(dco, co.data(), co.size() * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                 // `co` is a temporary
        // spline coefficients of the view: copy it out contiguously, filter the three axes in place
        const i64 tot = n[0] * n[1] * n[2];
        {
            // generic strided copy through k_trim_copy when the view is the [1:-1]^3 interior, else plain copy
            if (off == 0) HIP_TRY(hipMemcpyAsync(scratch, values, (size_t)tot * sizeof(T), hipMemcpyDeviceToDevice, st));
            else hipLaunchKernelGGL(k_trim_copy<T>, dim3((unsigned)std::min<i64>((tot + 255) / 256, 8192)), dim3(256), 0, st,
                                    scratch, values, n[0] + 2, n[1] + 2, n[2] + 2);
        }
        for (int a = 0; a < 3; ++a) {
            const i64 nl = tot / n[a];
            hipLaunchKernelGGL(k_spline_filter_axis<T>, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, st, scratch, n[0], n[1], n[2], a,
                               (method == 3 || method == 4) ? 1 : 0);
        }
        hipLaunchKernelGGL(k_spline_eval<T>, dim3(blocks), dim3(EMG_RCV_BLOCK), 0, st, out_dev, (const T*)scratch, n[0], n[1], n[2],
                           (const double*)dco, fac_host ? (const double*)dfac : nullptr, npts, cval, method == 3 ? 1 : method == 2 ? 2 : method == 4 ? 3 : 0);
    } else {
        std::vector<int> ii((size_t)3 * npts), ins((size_t)npts, 1);
        std::vector<double> tt((size_t)3 * npts);
        for (int a = 0; a < 3; ++a) {
            const double* g = pts[a].data();
            for (i64 r = 0; r < npts; ++r) {
                const double v = xi[a * npts + r];
                if (!(v >= g[0] && v <= g[n[a] - 1])) ins[r] = has_fill ? 0 : ins[r];
                i64 i = (i64)(std::lower_bound(g, g + n[a], v) - g) - 1;
                if (i < 0) i = 0;
                if (i > n[a] - 2) i = n[a] - 2;
                if (n[a] == 1) {
                    // an axis with ONE point (a 3-cell grid dimension, trimmed): RegularGridInterpolator's interval has no
                    // length, its normalised distance is 0 and both corners are that point (the stride is zeroed below)
                    ii[a * npts + r] = 0;
                    tt[a * npts + r] = 0.0;
                    continue;
                }
                ii[a * npts + r] = (int)i;
                tt[a * npts + r] = (v - g[i]) / (g[i + 1] - g[i]);
            }
        }
        HIP_TRY(hipMemcpyAsync(dco, tt.data(), tt.size() * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dii, ii.data(), ii.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dins, ins.data(), ins.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n[0] == 1) s0 = 0;
        if (n[1] == 1) s1 = 0;
        if (n[2] == 1) s2 = 0;
        hipLaunchKernelGGL(k_linear_eval<T>, dim3(blocks), dim3(EMG_RCV_BLOCK), 0, st, out_dev, values + off, s0, s1, s2,
                           (const int*)dii, (const double*)dco, (const int*)dins, fac_host ? (const double*)dfac : nullptr, npts, fill);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { fprintf(stderr, "[emg3d_hip] interp3d: %s\n", hipGetErrorString(e)); rc = (int)e; }
    return rc;
}

// fields.get_receiver_response (fields.py:733-817) of three device-resident components: resp_host[n] =
// sum_c fac[c][r] * interp3d(points_c[1:-1], comp_c[1:-1,1:-1,1:-1], xyz, 'cubic', 0.0, 'constant', nan); a component
// whose factors are all <= 1e-10 in magnitude is skipped (fields.py:810).
template <class T>
int receiver_response_device(hipStream_t st, const RcvComp<T> comp[3], i64 npts, const double* xyz, const double* fac,
                             T* scratch, T* resp_host) {
    if (npts < 1) return -2;
    T* dresp = nullptr;
    DEV_ALLOC(dresp, (size_t)npts * sizeof(T));
    HIP_TRY(hipMemsetAsync(dresp, 0, (size_t)npts * sizeof(T), st));
    int rc = 0;
    for (int c = 0; c < 3 && rc == 0; ++c) {
        bool active = false;
        for (i64 r = 0; r < npts; ++r) if (std::fabs(fac[c * npts + r]) > 1e-10) { active = true; break; }
        if (!active) continue;
        const RcvComp<T>& C = comp[c];
        i64 m[3];
        std::vector<double> p[3];
        for (int a = 0; a < 3; ++a) {
            m[a] = C.n[a] - 2;
            if (m[a] < 1) { rc = -2; break; }
            p[a].assign(C.pts[a].begin() + 1, C.pts[a].end() - 1);
        }
        if (rc) break;
        const i64 off = 1 + C.n[0] + C.n[0] * C.n[1];
        rc = interp3d_device<T>(st, C.dev, m, off, 1, C.n[0], C.n[0] * C.n[1], p, npts, xyz, 1, true, 0.0, NAN,
                                fac + c * npts, scratch, dresp);
    }
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(resp_host, dresp, (size_t)npts * sizeof(T), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = (int)e;
    }
        return rc;
}
