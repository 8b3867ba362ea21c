// Source vector of a finite electric dipole, built in HBM: the adjoint of trilinear interpolation that
// fields.get_source_field uses (reference emg3d/fields.py:586-629, _finite_source_xyz 914-1010).  Every cell
// the dipole crosses receives the fraction of the dipole inside it, split between its four edges of each
// component by the bilinear weights of the segment's midpoint.  Only the cells inside the dipole's bounding box
// are touched, so the nE-sized source never exists on the host: a solve uploads 6 coordinates instead of
// 51-102 MB (128^3) -- the step in front of the multigrid path.
//
// Gather form: one thread per EDGE of the bounding box sums the contributions of its (up to four) adjacent
// cells in the order in which the reference's loops (iz, iy, ix ascending; fields.py:960-962) add them, so the
// result does not depend on thread scheduling and equals the reference's accumulation order.
#pragma once
#include "common.hpp"

struct DipoleArgs {
    double src[6];            // x0, x1, y0, y1, z0, z1 (rounded to `decimals`)
    const double* nodes[3];   // rounded node coordinates (device)
    const double* h[3];       // cell widths (device, not rounded: fields.py:977-982)
    i64 nC[3];
    int lo[3], hi[3];         // loop ranges of the reference: cells lo[a] .. hi[a]-1
    FieldLayout fl;
};

// contribution of cell (ix, iy, iz): fraction of the dipole inside it and the relative position of the
// segment's midpoint (fields.py:964-985); returns false when the segment is not inside the cell
__device__ __forceinline__ bool dipole_cell(const DipoleArgs& a, int ix, int iy, int iz, double& x_len, double r[3]) {
    const int ic[3] = {ix, iy, iz};
    double d[3], al = 0.0, ar = 1.0;
    bool any = false;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        d[q] = a.src[2 * q + 1] - a.src[2 * q];
        if (d[q] != 0.0) {
            const double id = 1.0 / d[q];
            double a0 = (a.nodes[q][ic[q]] - a.src[2 * q]) * id, a1 = (a.nodes[q][ic[q] + 1] - a.src[2 * q]) * id;
            if (a0 > a1) { const double t = a0; a0 = a1; a1 = t; }
            al = any ? (a0 > al ? a0 : al) : a0;
            ar = any ? (a1 < ar ? a1 : ar) : a1;
            any = true;
        }
    }
    al = al > 0.0 ? al : 0.0;
    ar = ar < 1.0 ? ar : 1.0;
    double xc[3], dl2 = 0.0, sl2 = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double xmin = a.src[2 * q] + al * d[q], xmax = a.src[2 * q] + ar * d[q];
        xc[q] = (xmin + xmax) / 2.0;
        dl2 += (xmax - xmin) * (xmax - xmin);
        sl2 += d[q] * d[q];
        r[q] = (xc[q] - a.nodes[q][ic[q]]) / a.h[q][ic[q]];
    }
    x_len = sqrt(dl2) / sqrt(sl2);
    const double rmin = r[0] < r[1] ? (r[0] < r[2] ? r[0] : r[2]) : (r[1] < r[2] ? r[1] : r[2]);
    return rmin >= 0.0 && fabs(ar - al) > 0.0;
}

// component c of the source: s[edge] (+)= scale * weight; sums[block] = the block's weight (for the unity check, fields.py:1003).
// write == 0: only the sums.
// write == 1: divisor != 1 normalises the weights first (`s /= sum_s`, fields.py:1010).
template <class T>
__global__ void k_source_dipole(DipoleArgs a, int c, T* s, T scale, double* sums, int write, double divisor) {
    const int t1 = (c == 0) ? 1 : 0, t2 = (c == 2) ? 1 : 2;        // the two transverse axes, t1 < t2
    const int nc = a.hi[c] - a.lo[c], n1 = a.hi[t1] - a.lo[t1] + 1, n2 = a.hi[t2] - a.lo[t2] + 1;
    const int tot = nc * n1 * n2;
    double w = 0.0;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < tot && nc > 0) {
        int e[3];
        // edge index: along c a cell index, along t1 / t2 a node index
        const int ec = a.lo[c] + idx % nc, e1 = a.lo[t1] + (idx / nc) % n1, e2 = a.lo[t2] + idx / (nc * n1);
        e[c] = ec; e[t1] = e1; e[t2] = e2;
        // adjacent cells in the reference's loop order (t2 outer, t1 inner): (e1-1, e2-1), (e1, e2-1), (e1-1, e2), (e1, e2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b1 = (k & 1) ? 0 : 1, b2 = (k & 2) ? 0 : 1;   // 1: the cell BELOW the edge along that axis
            int ic[3];
            ic[c] = ec; ic[t1] = e1 - b1; ic[t2] = e2 - b2;
            if (ic[t1] < a.lo[t1] || ic[t1] >= a.hi[t1] || ic[t2] < a.lo[t2] || ic[t2] >= a.hi[t2]) continue;
            double x_len, r[3];
            if (!dipole_cell(a, ic[0], ic[1], ic[2], x_len, r)) continue;
            const double w1 = b1 ? r[t1] : 1.0 - r[t1], w2 = b2 ? r[t2] : 1.0 - r[t2];
            w += w1 * w2 * x_len;
        }
        if (write && w != 0.0) {
            const i64 off = a.fl.off[c] + e[0] * a.fl.st[c][0] + e[1] * a.fl.st[c][1] + e[2] * a.fl.st[c][2];
            s[off] += scale * (divisor == 1.0 ? w : w / divisor);
        }
    }
    if (!write) {
        // block sum -> one partial per block, added up by the host in block order (deterministic)
        __shared__ double red[256];
        red[threadIdx.x] = w;
        __syncthreads();
        for (int st = blockDim.x / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
    }
}
