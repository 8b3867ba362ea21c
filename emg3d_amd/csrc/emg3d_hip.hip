// C ABI of libemg3d_hip.so (see include/emg3d_hip.h).  gfx950 only.
#include "../../include/emg3d_hip.h"

#include <cstring>
#include <new>

#include <chrono>
#include "mg.hpp"
#include "receivers.hpp"
#include "source.hpp"
#include "gradient.hpp"

#define EMG3D_HIP_VERSION EMG3D_HIP_ABI_VERSION      // include/emg3d_hip.h

namespace {

template <class T>
MG<T>* as(emg3d_mg_t* mg) { return static_cast<MG<T>*>(reinterpret_cast<emg3d_mg*>(mg)); }

// The stateless entry points (tier 1, receivers, interpolation, cell averages, source field) run on the CALLING
// THREAD'S current device (emg3d_hip_set_device / hipSetDevice / torch.cuda.set_device) and leave it unchanged:
// a rank of a multi-GPU run works on its own GPU without passing a device to every call.
static int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }

template <class T> static T scalar_of(double re, double im);
template <> double scalar_of<double>(double re, double) { return re; }
template <> c128 scalar_of<c128>(double re, double im) { return mk(re, im); }

// eta arrays given directly (sv == false) or as REAL sigma*V arrays to be scaled by smu0 on the device
template <class T>
int create_impl(emg3d_mg_t** out, int dtype, i64 nx, i64 ny, i64 nz, const double* hx, const double* hy,
                const double* hz, const double* origin, const void* eta_x, const void* eta_y,
                const void* eta_z, const double* zeta, int device, bool sv = false, double smu0_re = 0.0,
                double smu0_im = 0.0, const double* vol = nullptr, bool resistivity = false,
                const double* epsr = nullptr, double seps0 = 0.0) {
    if (nx < 2 || ny < 2 || nz < 2) return -2;
    // (a failing step gives the handle's blocks back before it returns)
#define CREATE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { delete m; (void)hipGetLastError(); return (int)_e; } } while (0)
    HIP_TRY(hipSetDevice(device));
    MG<T>* m = new (std::nothrow) MG<T>();
    if (!m) return -3;
    m->dtype = dtype;
    m->device = device;
    m->stream = DevicePool::get().take_stream(device);       // a pooled stream of a closed handle, or a new one
    if (!m->stream) { delete m; return (int)hipErrorOutOfMemory; }
    m->own_stream = true;
    if (origin) for (int a = 0; a < 3; ++a) m->origin[a] = origin[a];
    std::vector<double> hh[3];
    hh[0].assign(hx, hx + nx); hh[1].assign(hy, hy + ny); hh[2].assign(hz, hz + nz);
    m->lv0 = m->make_level(hh);
    Level<T>& L = *m->lv0;
    const i64 nC = nx * ny * nz;
    m->eta_alias[1] = (eta_y == eta_x) || eta_y == nullptr;
    m->eta_alias[2] = (eta_z == eta_x) || eta_z == nullptr;
    if (!sv) {
        L.eta[0] = m->upload((const T*)eta_x, nC);
        L.eta[1] = m->eta_alias[1] ? L.eta[0] : m->upload((const T*)eta_y, nC);
        L.eta[2] = m->eta_alias[2] ? L.eta[0] : m->upload((const T*)eta_z, nC);
    } else {
        // sigma*V -- or, with `vol`, sigma and V -- stay in HBM (3-4 x 8 B per cell): emg3d_mg_set_smu0 forms eta of
        // another frequency from them
        const void* src[3] = {eta_x, eta_y, eta_z};
        if (vol) {
            m->volw = m->template dalloc<double>(nC);
            CREATE_TRY(m->h2d(m->volw, vol, (size_t)nC * sizeof(double)));
            if (epsr) {
                m->epsr = m->template dalloc<double>(nC);
                CREATE_TRY(m->h2d(m->epsr, epsr, (size_t)nC * sizeof(double)));
                m->seps0 = seps0;
            }
        }
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && m->eta_alias[c]) { L.eta[c] = L.eta[0]; m->sv[c] = m->sv[0]; continue; }
            L.eta[c] = m->template dalloc<T>(nC);
            m->sv[c] = m->template dalloc<double>(nC);
            CREATE_TRY(m->h2d(m->sv[c], src[c], (size_t)nC * sizeof(double)));
            if (resistivity && !m->broken)        // the arrays hold rho: sigma = 1 / rho on the device
                hipLaunchKernelGGL(k_recip_inplace, dim3((unsigned)std::min<i64>((nC + EMG_BLOCK - 1) / EMG_BLOCK, 4096)),
                                   dim3(EMG_BLOCK), 0, m->stream, m->sv[c], nC);
        }
        m->form_eta(L, scalar_of<T>(smu0_re, smu0_im));
    }
    L.zeta = m->upload(zeta, nC);
    m->check_zeta();
    m->norms = m->template dalloc<double>(MG<T>::NORM_SLOTS);
    hipMemsetAsync(L.s, 0, (size_t)L.nE * sizeof(T), m->stream);
    hipMemsetAsync(L.e, 0, (size_t)L.nE * sizeof(T), m->stream);
    hipMemsetAsync(L.r, 0, (size_t)L.nE * sizeof(T), m->stream);
    // default clevel: as MGParameters.max_level with clevel = -1 (solver.py:1155-1173)
    int cl[3];
    const i64 n3[3] = {nx, ny, nz};
    for (int a = 0; a < 3; ++a) { cl[a] = 0; i64 n = n3[a]; while (n % 2 == 0 && n > 2) { ++cl[a]; n /= 2; } }
    m->clevel[0] = std::max(cl[0], std::max(cl[1], cl[2]));
    m->clevel[1] = std::max(cl[1], cl[2]);
    m->clevel[2] = std::max(cl[0], cl[2]);
    m->clevel[3] = std::max(cl[0], cl[1]);
    if (hipStreamSynchronize(m->stream) != hipSuccess && m->err == 0) m->err = (int)hipGetLastError();
    if (m->broken && m->err == 0) m->err = (int)hipErrorOutOfMemory;
    // (the failed calls of a handle that could not get its memory leave HIP's per-thread "last error" set: whoever shares the
    // runtime -- torch -- would report it as its own at its next call)
    if (m->err) { int e = m->err; delete m; (void)hipGetLastError(); return e; }
    *out = reinterpret_cast<emg3d_mg_t*>(static_cast<emg3d_mg*>(m));
    return 0;
}
#undef CREATE_TRY

template <class T>
int finish(MG<T>* m) {
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->broken) { (void)hipGetLastError(); return (int)hipErrorOutOfMemory; }     // (a device allocation failed: the handle can only be destroyed)
    int e = m->err;
    m->err = 0;
    if (e) (void)hipGetLastError();         // (reported here: not left behind for the next user of the runtime, e.g. torch)
    return e;
}

#define DISPATCH(mg, CALL)                                             \
    do {                                                               \
        if (!mg) return -1;                                            \
        emg3d_mg* _b = reinterpret_cast<emg3d_mg*>(mg);                \
        if (_b->dtype) { typedef c128 T; MG<T>* m = as<T>(mg); if (m->broken) return (int)hipErrorOutOfMemory; CALL; } \
        else { typedef double T; MG<T>* m = as<T>(mg); if (m->broken) return (int)hipErrorOutOfMemory; CALL; }         \
    } while (0)

template <class T>
int set_field(MG<T>* m, T* dst, const void* host) {
    HIP_TRY(hipSetDevice(m->device));
    if (host) HIP_TRY(m->h2d(dst, host, (size_t)m->lv0->nE * sizeof(T)));
    else { HIP_TRY(hipMemsetAsync(dst, 0, (size_t)m->lv0->nE * sizeof(T), m->stream)); HIP_TRY(hipStreamSynchronize(m->stream)); }
    return 0;
}

template <class T>
int get_field(MG<T>* m, const T* src, void* host) {
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(m->d2h(host, src, (size_t)m->lv0->nE * sizeof(T)));
    return 0;
}

template <class T>
int read_norms(MG<T>* m, int n, double* out) {
    HIP_TRY(hipMemcpyAsync(out, m->norms, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    return finish(m);
}

// H = -curl E / smu0 (fields.py:819-911) between device buffers; see k_hfield.
template <class T>
void launch_hfield(hipStream_t stream, const i64 nC[3], const FieldLayout& fl, const T* e, const double* zeta,
                   double* const h[3], double* const ih[3], double sr, double si, T* out) {
    HFieldArgs<T> a;
    for (int q = 0; q < 3; ++q) { a.nC[q] = nC[q]; a.h[q] = h[q]; a.ih[q] = ih[q]; }
    a.fl = fl; a.e = e; a.zeta = zeta; a.out = out;
    if (sizeof(T) == 16) {      // Smith's complex division as NumPy evaluates it, constants formed once
        a.hi = std::fabs(sr) >= std::fabs(si);
        a.rat = a.hi ? si / sr : sr / si;
        a.scl = a.hi ? 1.0 / (sr + si * a.rat) : 1.0 / (si + sr * a.rat);
    } else { a.hi = 1; a.rat = 0.0; a.scl = sr; }
    const i64 plane = (nC[0] + 1) * (nC[1] + 1);
    dim3 grid((unsigned)((plane + EMG_BLOCK - 1) / EMG_BLOCK), (unsigned)(nC[2] + 1));
    hipLaunchKernelGGL(k_hfield<T>, grid, dim3(EMG_BLOCK), 0, stream, a);
}

static i64 hfield_size(const i64 nC[3]) {
    return (nC[0] + 1) * nC[1] * nC[2] + nC[0] * (nC[1] + 1) * nC[2] + nC[0] * nC[1] * (nC[2] + 1);
}

template <class T>
int hfield_impl(i64 nx, i64 ny, i64 nz, void* hf, const void* e, const double* zeta, const double* hx,
                const double* hy, const double* hz, double sr, double si) {
    const i64 nC[3] = {nx, ny, nz};
    const i64 nE = nx * (ny + 1) * (nz + 1) + (nx + 1) * ny * (nz + 1) + (nx + 1) * (ny + 1) * nz;
    const i64 nH = hfield_size(nC), ncell = nx * ny * nz;
    const double* hh[3] = {hx, hy, hz};
    // one device block: e | out | zeta | h, 1/h per axis
    const size_t bytes = (size_t)(nE + nH) * sizeof(T) + (size_t)(ncell + 2 * (nx + ny + nz)) * sizeof(double);
    char* base = nullptr;
    DEV_ALLOC(base, bytes);
    T* de = (T*)base;
    T* dout = de + nE;
    double* dz = (double*)(dout + nH);
    double* dh[3];
    double* dih[3];
    double* p = dz + ncell;
    std::vector<double> inv;
    hipError_t st = hipMemcpy(de, e, (size_t)nE * sizeof(T), hipMemcpyHostToDevice);
    if (st == hipSuccess && zeta) st = hipMemcpy(dz, zeta, (size_t)ncell * sizeof(double), hipMemcpyHostToDevice);
    for (int q = 0; q < 3 && st == hipSuccess; ++q) {
        dh[q] = p; dih[q] = p + nC[q]; p += 2 * nC[q];
        inv.assign(hh[q], hh[q] + nC[q]);
        for (double& v : inv) v = 1.0 / v;
        st = hipMemcpy(dh[q], hh[q], (size_t)nC[q] * sizeof(double), hipMemcpyHostToDevice);
        if (st == hipSuccess) st = hipMemcpy(dih[q], inv.data(), (size_t)nC[q] * sizeof(double), hipMemcpyHostToDevice);
    }
    if (st == hipSuccess) {
        launch_hfield<T>(nullptr, nC, ref_field_layout(nC), de, zeta ? dz : nullptr, dh, dih, sr, si, dout);
        st = hipGetLastError();
    }
    if (st == hipSuccess) st = hipMemcpy(hf, dout, (size_t)nH * sizeof(T), hipMemcpyDeviceToHost);
        if (st != hipSuccess) { fprintf(stderr, "[emg3d_hip] get_h_field: %s\n", hipGetErrorString(st)); return (int)st; }
    return 0;
}

// ------------------------------------------------------------------ tier 1
template <class T>
int amat_x_impl(i64 nx, i64 ny, i64 nz, void* r, const void* e, const void* ex, const void* ey,
                const void* ez, const double* zeta, const double* hx, const double* hy, const double* hz) {
    emg3d_mg_t* h = nullptr;
    int st = create_impl<T>(&h, sizeof(T) == 16, nx, ny, nz, hx, hy, hz, nullptr, ex, ey, ez, zeta, current_device());
    if (st) return st;
    MG<T>* m = as<T>(h);
    Level<T>& L = *m->lv0;
    st = set_field(m, L.e, e);
    if (!st) st = set_field(m, L.r, r);
    if (!st) {
        ResidualArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = L.nC[q]; a.eta[q] = L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q]; }
        a.fl = L.fl; a.r = L.r; a.s = L.r; a.e = L.e; a.zeta = L.zeta; a.partials = nullptr;
        const i64 plane = (nx + 1) * (ny + 1);
        dim3 grid((unsigned)((plane + EMG_BLOCK - 1) / EMG_BLOCK), (unsigned)(nz + 1));
        residual_launch<T>(0, 1, grid, m->stream, a);
        m->check_launch();
        st = finish(m);
    }
    if (!st) st = get_field(m, L.r, r);
    delete m;
    return st;
}

template <class T>
int gs_impl(int dir, i64 nx, i64 ny, i64 nz, void* e, const void* s, const void* ex, const void* ey,
            const void* ez, const double* zeta, const double* hx, const double* hy, const double* hz,
            int nu, int order) {
    if (dir < 0 || dir > 3 || (order != 0 && order != 1)) return -2;
    emg3d_mg_t* h = nullptr;
    int st = create_impl<T>(&h, sizeof(T) == 16, nx, ny, nz, hx, hy, hz, nullptr, ex, ey, ez, zeta, current_device());
    if (st) return st;
    MG<T>* m = as<T>(h);
    Level<T>& L = *m->lv0;
    m->order = order;
    m->use_home = 0;            // the field is read back from L.e right after the sweeps
    st = set_field(m, L.e, e);
    if (!st) st = set_field(m, L.s, s);
    if (!st) {
        if (dir == 0) m->smooth_point(L, nu);
        else m->smooth_line(L, dir - 1, nu);
        st = finish(m);
    }
    if (!st) st = get_field(m, L.e, e);
    delete m;
    return st;
}

template <class T>
int restrict_impl(i64 nx, i64 ny, i64 nz, i64 cnx, i64 cny, i64 cnz, void* cr, const void* r,
                  const double* const* w, int sc_dir) {
    if (sc_dir < 0 || sc_dir > 6) return -2;
    const i64 fn[3] = {nx, ny, nz}, cn[3] = {cnx, cny, cnz};
    int co[3];
    sc_axes(sc_dir, co);
    for (int a = 0; a < 3; ++a) if (cn[a] != (co[a] ? fn[a] / 2 : fn[a])) return -2;
    const i64 nEf = n_edges(fn), nEc = n_edges(cn);
    DevBlock br, bc, bw[9];
    double* dw[9] = {nullptr};
    HIP_TRY(br.alloc((size_t)nEf * sizeof(T)));
    HIP_TRY(bc.alloc((size_t)nEc * sizeof(T)));
    T *dr = br.get<T>(), *dc = bc.get<T>();
    HIP_TRY(hipMemcpy(dr, r, (size_t)nEf * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc, cr, (size_t)nEc * sizeof(T), hipMemcpyHostToDevice));
    RestrictArgs<T> a;
    for (int q = 0; q < 3; ++q) { a.cnC[q] = cn[q]; a.fnC[q] = fn[q]; a.co[q] = co[q]; }
    a.cfl = ref_field_layout(cn); a.ffl = ref_field_layout(fn); a.cr = dc; a.r = dr; a.pec = 0;
    for (int ax = 0; ax < 3; ++ax)
        for (int q = 0; q < 3; ++q) {
            a.w[ax][q] = nullptr;
            if (co[ax]) {
                const i64 n = cn[ax] + 1;
                HIP_TRY(bw[3 * ax + q].alloc((size_t)n * sizeof(double)));
                dw[3 * ax + q] = bw[3 * ax + q].get<double>();
                HIP_TRY(hipMemcpy(dw[3 * ax + q], w[3 * ax + q], (size_t)n * sizeof(double), hipMemcpyHostToDevice));
                a.w[ax][q] = dw[3 * ax + q];
            }
        }
    a.ce = nullptr;
    i64 nmax = 0;
    for (int c = 0; c < 3; ++c) {
        i64 n = 1;
        for (int q = 0; q < 3; ++q) n *= (q == c) ? cn[q] : cn[q] + 1;
        nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(k_restrict<T>, dim3((unsigned)((nmax + EMG_BLOCK - 1) / EMG_BLOCK), 3), dim3(EMG_BLOCK), 0, 0, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cr, dc, (size_t)nEc * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <class T>
int prolong_impl(i64 nx, i64 ny, i64 nz, const double* hx, const double* hy, const double* hz,
                 const double* origin, void* e, const void* ce, int sc_dir) {
    if (sc_dir < 0 || sc_dir > 6) return -2;
    // a throw-away two-level hierarchy with unit model (only grids/weights are used)
    const i64 nC = nx * ny * nz;
    std::vector<T> eta((size_t)nC);
    std::vector<double> zeta((size_t)nC, 1.0);
    memset(eta.data(), 0, sizeof(T) * (size_t)nC);
    emg3d_mg_t* h = nullptr;
    int st = create_impl<T>(&h, sizeof(T) == 16, nx, ny, nz, hx, hy, hz, origin, eta.data(), eta.data(),
                            eta.data(), zeta.data(), current_device());
    if (st) return st;
    MG<T>* m = as<T>(h);
    Level<T>& L = *m->lv0;
    Transfer X;
    auto C = m->make_child(L, X, sc_dir);
    st = set_field(m, L.e, e);
    if (!st) {
        HIP_TRY(hipMemcpyAsync(C->e, ce, (size_t)C->nE * sizeof(T), hipMemcpyHostToDevice, m->stream));
        m->prolong_from(L, X, *C);
        st = finish(m);
    }
    if (!st) st = get_field(m, L.e, e);
    delete m;
    return st;
}

template <class T>
int restrict_model_impl(i64 nx, i64 ny, i64 nz, void* cp, const void* p, int sc_dir) {
    if (sc_dir < 0 || sc_dir > 6) return -2;
    int co[3];
    sc_axes(sc_dir, co);
    const i64 cnx = co[0] ? nx / 2 : nx, cny = co[1] ? ny / 2 : ny, cnz = co[2] ? nz / 2 : nz;
    const i64 nf = nx * ny * nz, nc = cnx * cny * cnz;
    DevBlock bp, bc;
    HIP_TRY(bp.alloc((size_t)nf * sizeof(T)));
    HIP_TRY(bc.alloc((size_t)nc * sizeof(T)));
    T *dp = bp.get<T>(), *dc = bc.get<T>();
    HIP_TRY(hipMemcpy(dp, p, (size_t)nf * sizeof(T), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_restrict_model<T>, dim3((unsigned)((nc + EMG_BLOCK - 1) / EMG_BLOCK)), dim3(EMG_BLOCK), 0, 0,
                       dc, (const T*)dp, cnx, cny, cnz, nx, ny, co[0], co[1], co[2]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cp, dc, (size_t)nc * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <class T>
int solve_impl(void* amat, void* bvec, i64 n) {
    DevBlock ba, bb;
    HIP_TRY(ba.alloc((size_t)(6 * n) * sizeof(T)));
    HIP_TRY(bb.alloc((size_t)n * sizeof(T)));
    T *da = ba.get<T>(), *db = bb.get<T>();
    HIP_TRY(hipMemcpy(da, amat, (size_t)(6 * n) * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, bvec, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_solve_banded<T>, dim3(1), dim3(1), 0, 0, da, db, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(amat, da, (size_t)(6 * n) * sizeof(T), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(bvec, db, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <class T>
int b2a_impl(void* amat, void* bvec, i64 n, const void* middle, const double* left, const void* rhs, i64 im, i64 nC) {
    T *da = nullptr, *db = nullptr, *dm = nullptr, *dr = nullptr;
    double* dl = nullptr;
    DEV_ALLOC(da, (size_t)(6 * n) * sizeof(T));
    DEV_ALLOC(db, (size_t)n * sizeof(T));
    DEV_ALLOC(dm, 25 * sizeof(T));
    DEV_ALLOC(dr, 5 * sizeof(T));
    DEV_ALLOC(dl, 25 * sizeof(double));
    HIP_TRY(hipMemcpy(da, amat, (size_t)(6 * n) * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, bvec, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dm, middle, 25 * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr, rhs, 5 * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dl, left, 25 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_blocks_to_amat<T>, dim3(1), dim3(1), 0, 0, da, db, (const T*)dm, (const double*)dl, (const T*)dr, im, nC);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(amat, da, (size_t)(6 * n) * sizeof(T), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(bvec, db, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
        return 0;
}

// fields.get_source_field for ONE finite dipole segment (fields.py:586-629, 914-1010) into a device field:
// s (+)= scale[c] * weights_c.  nodes = unrounded node vectors (host), h = device cell widths.  Returns -4 when the
// source lies outside the grid (the reference raises ValueError).  sums3: the three component sums before scaling.
template <class T>
int source_dipole_device(hipStream_t st, const std::vector<double> nodes[3], double* const hdev[3], const i64 nC[3],
                         const FieldLayout& fl, const double* src6, const double* scale6, int decimals, T* s, double* sums3) {
    const double p10 = std::pow(10.0, decimals);
    auto rnd = [&](double v) { return std::nearbyint(v * p10) / p10; };       // numpy.round(v, decimals)
    DipoleArgs a;
    std::vector<double> rn[3];
    for (int q = 0; q < 3; ++q) {
        rn[q].resize(nodes[q].size());
        for (size_t i = 0; i < nodes[q].size(); ++i) rn[q][i] = rnd(nodes[q][i]);
        a.src[2 * q] = rnd(src6[2 * q]); a.src[2 * q + 1] = rnd(src6[2 * q + 1]);
        a.nC[q] = nC[q]; a.h[q] = hdev[q];
        if (a.src[2 * q] < rn[q].front() || a.src[2 * q + 1] > rn[q].back()) return -4;     // fields.py:926-931
    }
    double len2 = 0.0;
    for (int q = 0; q < 3; ++q) len2 += (a.src[2 * q + 1] - a.src[2 * q]) * (a.src[2 * q + 1] - a.src[2 * q]);
    if (len2 == 0.0) return -2;
    for (int q = 0; q < 3; ++q) {       // min_max_ind, fields.py:947-952
        const double vmin = std::min(a.src[2 * q], a.src[2 * q + 1]), vmax = std::max(a.src[2 * q], a.src[2 * q + 1]);
        const i64 i0 = (i64)(std::upper_bound(rn[q].begin(), rn[q].end(), vmin) - rn[q].begin()) - 1;
        const i64 i1 = (i64)(std::upper_bound(rn[q].begin(), rn[q].end(), vmax) - rn[q].begin()) - 1;
        a.lo[q] = (int)std::max<i64>(0, i0);
        a.hi[q] = (int)std::min<i64>(std::max<i64>(0, i1) + 1, (i64)rn[q].size() - 1);
    }
    a.fl = fl;
    auto grid_of = [&](int c) {
        const int t1 = (c == 0) ? 1 : 0, t2 = (c == 2) ? 1 : 2;
        const i64 n = (i64)std::max(0, a.hi[c] - a.lo[c]) * (a.hi[t1] - a.lo[t1] + 1) * (a.hi[t2] - a.lo[t2] + 1);
        return (unsigned)std::max<i64>(1, (n + 255) / 256);
    };
    const size_t tot = rn[0].size() + rn[1].size() + rn[2].size();
    const size_t nbl[3] = {grid_of(0), grid_of(1), grid_of(2)};
    DevBlock bn;                                                     // rounded nodes | per-block partial sums
    HIP_TRY(bn.alloc((tot + nbl[0] + nbl[1] + nbl[2]) * sizeof(double)));
    double* dn = bn.get<double>();
    double* dpart[3] = {dn + tot, dn + tot + nbl[0], dn + tot + nbl[0] + nbl[1]};
    {
        std::vector<double> all;
        for (int q = 0; q < 3; ++q) all.insert(all.end(), rn[q].begin(), rn[q].end());
        HIP_TRY(hipMemcpyAsync(dn, all.data(), tot * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    a.nodes[0] = dn; a.nodes[1] = dn + rn[0].size(); a.nodes[2] = dn + rn[0].size() + rn[1].size();
    for (int c = 0; c < 3; ++c)
        hipLaunchKernelGGL(k_source_dipole<T>, dim3(grid_of(c)), dim3(256), 0, st, a, c, s, Zero<T>::v(), dpart[c], 0, 1.0);
    std::vector<double> part(nbl[0] + nbl[1] + nbl[2]);
    HIP_TRY(hipMemcpyAsync(part.data(), dpart[0], part.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    size_t po = 0;
    for (int c = 0; c < 3; ++c) {
        double sum = 0.0;
        for (size_t b = 0; b < nbl[c]; ++b) sum += part[po + b];     // fixed order
        po += nbl[c];
        if (sums3) sums3[c] = sum;
        if (a.src[2 * c + 1] - a.src[2 * c] == 0.0) continue;        // no moment along this axis
        const T sc = scalar_of<T>(scale6[2 * c], scale6[2 * c + 1]);
        // "Normalizing Source", fields.py:1003-1010: s /= |sum| whenever |sum| differs from one by more than 1e-6 (the
        // caller reports it: the sums go back through sums3, emg3d_amd/solver.py raises the reference's warning)
        const double ss = std::fabs(sum);
        const double divisor = (std::fabs(ss - 1.0) > 1e-6) ? ss : 1.0;
        hipLaunchKernelGGL(k_source_dipole<T>, dim3(grid_of(c)), dim3(256), 0, st, a, c, s, sc, dpart[c], 1, divisor);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e == hipSuccess ? 0 : (int)e;
}

// grid vectors of the three field components (electric: edges, magnetic: faces), fields.py:787-797
template <class T>
void receiver_components(const std::vector<double> nodes[3], const std::vector<double> centers[3], const i64 nC[3],
                         bool electric, const T* field_dev, RcvComp<T> comp[3]) {
    i64 o = 0;
    for (int c = 0; c < 3; ++c) {
        comp[c].dev = field_dev + o;
        i64 tot = 1;
        for (int a = 0; a < 3; ++a) {
            const bool cell = electric ? (a == c) : (a != c);      // cell-centred along this axis?
            comp[c].n[a] = cell ? nC[a] : nC[a] + 1;
            comp[c].pts[a] = cell ? centers[a] : nodes[a];
            tot *= comp[c].n[a];
        }
        o += tot;
    }
}

inline void grid_vectors(const double* h, i64 n, double origin, std::vector<double>& nodes, std::vector<double>& centers) {
    nodes.resize(n + 1); centers.resize(n);
    double cs = 0.0;
    nodes[0] = 0.0 + origin;
    for (i64 i = 0; i < n; ++i) { cs += h[i]; nodes[i + 1] = cs + origin; }        // np.r_[0, cumsum(h)] + origin
    for (i64 i = 0; i < n; ++i) centers[i] = (nodes[i + 1] + nodes[i]) / 2;
}

template <class T>
int receiver_host_impl(i64 nx, i64 ny, i64 nz, const double* hx, const double* hy, const double* hz, const double* origin,
                       const void* field, int is_electric, i64 n, const double* xyz, const double* fac, void* resp) {
    const i64 nC[3] = {nx, ny, nz};
    const double* hh[3] = {hx, hy, hz};
    std::vector<double> nodes[3], centers[3];
    for (int a = 0; a < 3; ++a) grid_vectors(hh[a], nC[a], origin ? origin[a] : 0.0, nodes[a], centers[a]);
    const i64 nF = is_electric ? n_edges(nC) : hfield_size(nC);
    T *df = nullptr, *scr = nullptr;
    DEV_ALLOC(df, (size_t)nF * sizeof(T));
    DEV_ALLOC(scr, (size_t)nF * sizeof(T));
    HIP_TRY(hipMemcpy(df, field, (size_t)nF * sizeof(T), hipMemcpyHostToDevice));
    RcvComp<T> comp[3];
    receiver_components<T>(nodes, centers, nC, is_electric != 0, df, comp);
    const int rc = receiver_response_device<T>(nullptr, comp, n, xyz, fac, scr, (T*)resp);
        return rc;
}

template <class T>
int interp3d_host_impl(i64 nx, i64 ny, i64 nz, const double* px, const double* py, const double* pz, const void* values,
                       i64 n, const double* xi, int method, int has_fill, double fill, double cval, void* out) {
    const i64 nn[3] = {nx, ny, nz};
    std::vector<double> pts[3];
    pts[0].assign(px, px + nx); pts[1].assign(py, py + ny); pts[2].assign(pz, pz + nz);
    const i64 tot = nx * ny * nz;
    T *dv = nullptr, *scr = nullptr, *dout = nullptr;
    DEV_ALLOC(dv, (size_t)tot * sizeof(T));
    DEV_ALLOC(scr, (size_t)tot * sizeof(T));
    DEV_ALLOC(dout, (size_t)n * sizeof(T));
    HIP_TRY(hipMemcpy(dv, values, (size_t)tot * sizeof(T), hipMemcpyHostToDevice));
    int rc = interp3d_device<T>(nullptr, dv, nn, 0, 1, nx, nx * ny, pts, n, xi, method, has_fill != 0, fill, cval, nullptr, scr, dout);
    if (rc == 0 && hipMemcpy(out, dout, (size_t)n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) rc = (int)hipGetLastError();
        return rc;
}

}  // namespace

// The field in the reference layout.  On levels that keep the field in the x-split working copy between cycles (MG::home_on)
// this converts it back first (kernels on the handle's stream): the pointer is a SNAPSHOT, valid until the next cycle or
// smoothing call on the handle -- fetch it again afterwards (emg3d_amd.shard.efield_tensor does).
template <class T>
static void* efield_ptr(MG<T>* m) {
    if (hipSetDevice(m->device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    T* p = m->sel_e();
    m->check_launch();
    return m->err ? nullptr : (void*)p;
}

template <class T>
static int sweep_plan_impl(i64 nx, i64 ny, i64 nz, int dir, int order, int nsys, int cu_count, char* name, int64_t* info) {
    MG<T> m;
    if (cu_count > 0) {                     // a device of that size, whatever this machine has
        m.cu_count = cu_count;
        m.tha_lds_state = 1; m.tha_lds_limit = 160 * 1024;
    } else m.device = current_device();
    m.order = order; m.nsys = nsys;
    m.lv0 = std::make_shared<Level<T>>();
    Level<T>& L = *m.lv0;
    L.nC[0] = nx; L.nC[1] = ny; L.nC[2] = nz;
    MG<T>::shape_level(L);
    i64 inf[6];
    m.plan_sweep(L, dir - 1, name, inf);
    for (int k = 0; k < 6; ++k) info[k] = inf[k];
    return 0;
}

extern "C" {

int emg3d_hip_version(void) { return EMG3D_HIP_VERSION; }

int64_t emg3d_hip_release_cached(void) { return (int64_t)DevicePool::get().release_all(); }
int64_t emg3d_hip_cached_bytes(void) { return (int64_t)DevicePool::get().bytes_held(); }
int64_t emg3d_hip_cached_bytes_on(int device) { return device < 0 ? 0 : (int64_t)DevicePool::get().bytes_held_on(device); }

int emg3d_hip_device_count(int* count) {
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

int emg3d_hip_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int emg3d_hip_device_info(int device, char* name, int64_t* total_mem, int* cu_count) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name) { strncpy(name, p.name, 255); name[255] = 0; }
    if (total_mem) *total_mem = (int64_t)p.totalGlobalMem;
    if (cu_count) *cu_count = p.multiProcessorCount;
    return 0;
}

int emg3d_hip_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes) {
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    size_t fr = 0, tot = 0;
    const hipError_t e = hipMemGetInfo(&fr, &tot);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    if (free_bytes) *free_bytes = (int64_t)fr;
    if (total_bytes) *total_bytes = (int64_t)tot;
    return 0;
}

int emg3d_amat_x(int dtype, int64_t nx, int64_t ny, int64_t nz, void* r, const void* e, const void* eta_x,
                 const void* eta_y, const void* eta_z, const double* zeta, const double* hx,
                 const double* hy, const double* hz) {
    return dtype ? amat_x_impl<c128>(nx, ny, nz, r, e, eta_x, eta_y, eta_z, zeta, hx, hy, hz)
                 : amat_x_impl<double>(nx, ny, nz, r, e, eta_x, eta_y, eta_z, zeta, hx, hy, hz);
}

int emg3d_get_h_field(int dtype, int64_t nx, int64_t ny, int64_t nz, void* hfield, const void* efield,
                      const double* zeta, const double* hx, const double* hy, const double* hz,
                      double smu0_re, double smu0_im) {
    if (nx < 1 || ny < 1 || nz < 1 || !hfield || !efield || !hx || !hy || !hz) return -2;
    if (smu0_re == 0.0 && smu0_im == 0.0) return -2;
    if (!dtype && smu0_im != 0.0) return -2;
    return dtype ? hfield_impl<c128>(nx, ny, nz, hfield, efield, zeta, hx, hy, hz, smu0_re, smu0_im)
                 : hfield_impl<double>(nx, ny, nz, hfield, efield, zeta, hx, hy, hz, smu0_re, smu0_im);
}

int emg3d_gauss_seidel(int dtype, int dir, int64_t nx, int64_t ny, int64_t nz, void* e, const void* s,
                       const void* eta_x, const void* eta_y, const void* eta_z, const double* zeta,
                       const double* hx, const double* hy, const double* hz, int nu, int order) {
    return dtype ? gs_impl<c128>(dir, nx, ny, nz, e, s, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu, order)
                 : gs_impl<double>(dir, nx, ny, nz, e, s, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu, order);
}

int emg3d_restrict(int dtype, int64_t nx, int64_t ny, int64_t nz, int64_t cnx, int64_t cny, int64_t cnz,
                   void* cr, const void* r, const double* const* w, int sc_dir) {
    return dtype ? restrict_impl<c128>(nx, ny, nz, cnx, cny, cnz, cr, r, w, sc_dir)
                 : restrict_impl<double>(nx, ny, nz, cnx, cny, cnz, cr, r, w, sc_dir);
}

int emg3d_restrict_weights(const double* vectorN, const double* vectorCC, const double* h, int64_t nh,
                           const double* cvectorN, const double* cvectorCC, const double* ch, int64_t n,
                           double* wl, double* w0, double* wr) {
    if (n < 2 || nh < 2) return -2;
    restrict_weights_host(vectorN, vectorCC, h, nh, cvectorN, cvectorCC, ch, n, wl, w0, wr);
    return 0;
}

int emg3d_solve(int dtype, void* amat, void* bvec, int64_t n) {
    if (n < 1) return -2;
    return dtype ? solve_impl<c128>(amat, bvec, n) : solve_impl<double>(amat, bvec, n);
}

int emg3d_blocks_to_amat(int dtype, void* amat, void* bvec, int64_t n, const void* middle, const double* left,
                         const void* rhs, int64_t im, int64_t nC) {
    return dtype ? b2a_impl<c128>(amat, bvec, n, middle, left, rhs, im, nC)
                 : b2a_impl<double>(amat, bvec, n, middle, left, rhs, im, nC);
}

int emg3d_prolongation(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx, const double* hy,
                       const double* hz, const double* origin, void* e, const void* ce, int sc_dir) {
    return dtype ? prolong_impl<c128>(nx, ny, nz, hx, hy, hz, origin, e, ce, sc_dir)
                 : prolong_impl<double>(nx, ny, nz, hx, hy, hz, origin, e, ce, sc_dir);
}

int emg3d_restrict_model(int is_complex, int64_t nx, int64_t ny, int64_t nz, void* cparam, const void* param,
                         int sc_dir) {
    return is_complex ? restrict_model_impl<c128>(nx, ny, nz, cparam, param, sc_dir)
                      : restrict_model_impl<double>(nx, ny, nz, cparam, param, sc_dir);
}

// ------------------------------------------------------------------ tier 2
int emg3d_sweep_plan(int dtype, int64_t nx, int64_t ny, int64_t nz, int dir, int order, int nsys, int cu_count, char* name,
                     int64_t* info) {
    if (nx < 2 || ny < 2 || nz < 2 || dir < 1 || dir > 3 || order < 0 || order > 1 || nsys < 1 || nsys > 64 || !name || !info) return -2;
    return dtype ? sweep_plan_impl<c128>(nx, ny, nz, dir, order, nsys, cu_count, name, info)
                 : sweep_plan_impl<double>(nx, ny, nz, dir, order, nsys, cu_count, name, info);
}

int emg3d_mg_create(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                    const double* hy, const double* hz, const double* origin, const void* eta_x,
                    const void* eta_y, const void* eta_z, const double* zeta, int device) {
    if (!out) return -1;
    return dtype ? create_impl<c128>(out, 1, nx, ny, nz, hx, hy, hz, origin, eta_x, eta_y, eta_z, zeta, device)
                 : create_impl<double>(out, 0, nx, ny, nz, hx, hy, hz, origin, eta_x, eta_y, eta_z, zeta, device);
}

int emg3d_mg_create_sv(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                       const double* hy, const double* hz, const double* origin, const double* sv_x,
                       const double* sv_y, const double* sv_z, const double* zeta, double smu0_re,
                       double smu0_im, int device) {
    if (!out || !sv_x) return -1;
    return dtype ? create_impl<c128>(out, 1, nx, ny, nz, hx, hy, hz, origin, sv_x, sv_y, sv_z, zeta, device, true, smu0_re, smu0_im)
                 : create_impl<double>(out, 0, nx, ny, nz, hx, hy, hz, origin, sv_x, sv_y, sv_z, zeta, device, true, smu0_re, smu0_im);
}

int emg3d_mg_create_vs(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                       const double* hy, const double* hz, const double* origin, const double* sigma_x,
                       const double* sigma_y, const double* sigma_z, const double* vol, const double* zeta,
                       double smu0_re, double smu0_im, int resistivity, int device) {
    if (!out || !sigma_x || !vol) return -1;
    if (dtype ? smu0_re != 0.0 : smu0_im != 0.0) return -2;     // i b (frequency domain) or real (Laplace domain)
    return dtype ? create_impl<c128>(out, 1, nx, ny, nz, hx, hy, hz, origin, sigma_x, sigma_y, sigma_z, zeta, device, true, smu0_re, smu0_im, vol, resistivity != 0)
                 : create_impl<double>(out, 0, nx, ny, nz, hx, hy, hz, origin, sigma_x, sigma_y, sigma_z, zeta, device, true, smu0_re, smu0_im, vol, resistivity != 0);
}

int emg3d_mg_create_vse(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                        const double* hy, const double* hz, const double* origin, const double* sigma_x,
                        const double* sigma_y, const double* sigma_z, const double* vol, const double* zeta,
                        const double* epsilon_r, double smu0_re, double smu0_im, double seps0, int resistivity, int device) {
    if (!out || !sigma_x || !vol || !epsilon_r) return -1;
    if (dtype ? smu0_re != 0.0 : smu0_im != 0.0) return -2;     // i b (frequency domain) or real (Laplace domain)
    return dtype ? create_impl<c128>(out, 1, nx, ny, nz, hx, hy, hz, origin, sigma_x, sigma_y, sigma_z, zeta, device, true, smu0_re, smu0_im, vol, resistivity != 0, epsilon_r, seps0)
                 : create_impl<double>(out, 0, nx, ny, nz, hx, hy, hz, origin, sigma_x, sigma_y, sigma_z, zeta, device, true, smu0_re, smu0_im, vol, resistivity != 0, epsilon_r, seps0);
}

void emg3d_mg_destroy(emg3d_mg_t* mg) {
    if (!mg) return;
    const bool broken = reinterpret_cast<emg3d_mg*>(mg)->dtype ? as<c128>(mg)->broken : as<double>(mg)->broken;
    delete reinterpret_cast<emg3d_mg*>(mg);
    if (broken) (void)hipGetLastError();        // (see finish())
}

int emg3d_mg_set_params(emg3d_mg_t* mg, int cycle, int nu_init, int nu_pre, int nu_coarse, int nu_post,
                        const int* clevel, int order) {
    if (cycle != 'V' && cycle != 'W' && cycle != 'F') return -2;
    if (order != 0 && order != 1) return -2;
    DISPATCH(mg, {
        // the captured launch sequences depend on every one of these: drop them only when something changes (a handle
        // that is re-used for the next frequency / source keeps its graphs)
        bool same = m->cycle == cycle && m->nu_init == nu_init && m->nu_pre == nu_pre && m->nu_coarse == nu_coarse &&
                    m->nu_post == nu_post && m->order == order;
        if (clevel) for (int q = 0; q < 4; ++q) same = same && m->clevel[q] == clevel[q];
        if (!same) m->drop_graphs();
        m->cycle = cycle; m->cycmax = (cycle == 'V') ? 1 : 2;
        m->entry_cm = 0;
        m->nu_init = nu_init; m->nu_pre = nu_pre; m->nu_coarse = nu_coarse; m->nu_post = nu_post;
        if (clevel) {
            bool changed = false;
            for (int q = 0; q < 4; ++q) { if (m->clevel[q] != clevel[q]) changed = true; m->clevel[q] = clevel[q]; }
            if (changed) m->hier.clear();   // device arrays stay allocated until destroy
        }
        if (m->order != order) {
            // the kernel (and with it the layout of the cached line factors) is chosen per ordering:
            // rebuild them on next use (the old device arrays stay allocated until destroy).  The field may live in
            // the x-split working copy of the colour ordering (MG::home_on): back to the reference layout first -- the
            // eager launch path of the other ordering reads L.e directly.
            HIP_TRY(hipSetDevice(m->device));
            m->e_to_ref(*m->lv0);
            m->check_launch();
            m->order = order;
            m->forget_factors();
        }
        return 0;
    });
}

int emg3d_mg_set_smu0(emg3d_mg_t* mg, double smu0_re, double smu0_im) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        if (sizeof(T) == 8 && smu0_im != 0.0) return -2;       // a float64 (Laplace-domain) handle takes a real s mu_0
        if (sizeof(T) == 16 && m->volw && smu0_re != 0.0) return -2;   // (sigma, V) handles: s mu_0 = i b
        if (m->epsr) return -7;                                        // eps_r handles: emg3d_mg_set_smu0_eps (needs s eps_0 too)
        const int st = m->set_smu0(scalar_of<T>(smu0_re, smu0_im));
        if (st) return st;
        return finish(m);
    });
}

int emg3d_mg_set_smu0_eps(emg3d_mg_t* mg, double smu0_re, double smu0_im, double seps0) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        if (!m->epsr) return -7;                                // not a handle with relative permittivities
        if (sizeof(T) == 8 && smu0_im != 0.0) return -2;
        if (sizeof(T) == 16 && smu0_re != 0.0) return -2;
        const double seps0_before = m->seps0;
        m->seps0 = seps0;
        const int st = m->set_smu0(scalar_of<T>(smu0_re, smu0_im));
        if (st) { m->seps0 = seps0_before; return st; }         // (the handle keeps a consistent (s mu_0, s eps_0) pair)
        return finish(m);
    });
}

int emg3d_mg_begin(emg3d_mg_t* mg, int sc_dir) {
    if (sc_dir < 0 || sc_dir > 3) return -2;
    DISPATCH(mg, { m->entry_cm = (0 == m->clevel[sc_dir]) ? 1 : m->cycmax; return 0; });
}

int emg3d_mg_set_sfield(emg3d_mg_t* mg, const void* s) { DISPATCH(mg, { m->source_changed(); return set_field(m, m->sel_s(), s); }); }
int emg3d_mg_set_sfield_vector(emg3d_mg_t* mg, const double* vector, double smu0_re, double smu0_im) {
    if (!vector) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        // stage the real vector in r (same size in bytes or larger), scale into s
        double* tmp = reinterpret_cast<double*>(L.r);
        HIP_TRY(m->h2d(tmp, vector, (size_t)L.nE * sizeof(double)));
        const unsigned blocks = (unsigned)std::min<i64>((L.nE + EMG_BLOCK - 1) / EMG_BLOCK, 4096);
        hipLaunchKernelGGL(k_scale_real_to<T>, dim3(blocks), dim3(EMG_BLOCK), 0, m->stream, m->sel_s(), (const double*)tmp,
                           scalar_of<T>(smu0_re, smu0_im), L.nE);
        m->source_changed();
        m->check_launch();
        return finish(m);
    });
}
int emg3d_edges2cellaverages(int dtype, int64_t nx, int64_t ny, int64_t nz, const void* field, const double* vol,
                             void* out_x, void* out_y, void* out_z) {
    if (nx < 1 || ny < 1 || nz < 1 || !field || !vol || !out_x || !out_y || !out_z) return -2;
    const i64 nC[3] = {nx, ny, nz};
    const i64 nE = n_edges(nC), n = nx * ny * nz;
    const size_t ts = dtype ? 16 : 8;
    char* base = nullptr;
    DEV_ALLOC(base, (size_t)(nE + 3 * n) * ts + (size_t)n * 8);
    char* df = base; char* dout = base + (size_t)nE * ts; double* dvol = (double*)(dout + (size_t)3 * n * ts);
    void* outs[3] = {out_x, out_y, out_z};
    HIP_TRY(hipMemcpy(df, field, (size_t)nE * ts, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dvol, vol, (size_t)n * 8, hipMemcpyHostToDevice));
    for (int c = 0; c < 3; ++c) HIP_TRY(hipMemcpy(dout + (size_t)c * n * ts, outs[c], (size_t)n * ts, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype) {
        E2CArgs<c128> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = nC[q]; a.h[q] = nullptr; a.out[q] = (c128*)dout + (size_t)q * n; }
        a.fl = ref_field_layout(nC); a.f = (const c128*)df; a.vol = dvol;
        hipLaunchKernelGGL(k_edges2cell<c128>, dim3(blocks), dim3(256), 0, 0, a);
    } else {
        E2CArgs<double> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = nC[q]; a.h[q] = nullptr; a.out[q] = (double*)dout + (size_t)q * n; }
        a.fl = ref_field_layout(nC); a.f = (const double*)df; a.vol = dvol;
        hipLaunchKernelGGL(k_edges2cell<double>, dim3(blocks), dim3(256), 0, 0, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    for (int c = 0; c < 3; ++c) HIP_TRY(hipMemcpy(outs[c], dout + (size_t)c * n * ts, (size_t)n * ts, hipMemcpyDeviceToHost));
        return 0;
}

int emg3d_mg_gradient(emg3d_mg_t* mg, int efield_vec, double smu0_re, double smu0_im, double* grad) {
    if (!mg || !grad) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        const T* fwd = m->vec(efield_vec);
        if (!fwd || efield_vec == -2) return -2;           // the forward field must be a saved copy, not the live field
        // the gradient (nC doubles) is staged in the residual buffer (nC * 8 < nE * sizeof(T))
        double* dg = reinterpret_cast<double*>(L.r);
        const unsigned blocks = (unsigned)((L.nCells + 255) / 256);
        hipLaunchKernelGGL(k_gradient<T>, dim3(blocks), dim3(256), 0, m->stream, L.nC[0], L.nC[1], L.nC[2], L.fl, fwd,
                           (const T*)m->sel_e(), smu0_re, smu0_im, (const double*)L.h[0], (const double*)L.h[1], (const double*)L.h[2], dg);
        m->check_launch();
        HIP_TRY(m->d2h(grad, dg, (size_t)L.nCells * sizeof(double)));
        return finish(m);
    });
}

int emg3d_mg_set_sfield_dipole(emg3d_mg_t* mg, const double* src6, const double* scale6, int decimals, int accumulate,
                               double* sums3) {
    if (!mg || !src6 || !scale6) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        if (!accumulate) HIP_TRY(hipMemsetAsync(m->sel_s(), 0, (size_t)L.nE * sizeof(T), m->stream));
        m->source_changed();
        const int rc = source_dipole_device<T>(m->stream, L.nodes, L.h, L.nC, L.fl, src6, scale6, decimals, m->sel_s(), sums3);
        const int st = finish(m);
        return rc ? rc : st;
    });
}

int emg3d_source_field(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx, const double* hy, const double* hz,
                       const double* origin, const double* src6, const double* scale6, int decimals, void* sfield,
                       double* sums3) {
    if (nx < 1 || ny < 1 || nz < 1 || !hx || !hy || !hz || !src6 || !scale6 || !sfield) return -2;
    emg3d_mg_t* h = nullptr;
    // a throw-away handle gives the device grid vectors (the model is not used)
    const i64 nC = nx * ny * nz;
    if (nx < 2 || ny < 2 || nz < 2) return -2;
    std::vector<double> zeta((size_t)nC, 1.0);
    int st;
    if (dtype) {
        std::vector<c128> eta((size_t)nC, mk(0.0, 0.0));
        st = create_impl<c128>(&h, 1, nx, ny, nz, hx, hy, hz, origin, eta.data(), eta.data(), eta.data(), zeta.data(), current_device());
    } else {
        std::vector<double> eta((size_t)nC, 0.0);
        st = create_impl<double>(&h, 0, nx, ny, nz, hx, hy, hz, origin, eta.data(), eta.data(), eta.data(), zeta.data(), current_device());
    }
    if (st) return st;
    st = emg3d_mg_set_sfield_dipole(h, src6, scale6, decimals, 0, sums3);
    if (!st) {
        if (dtype) st = get_field(as<c128>(h), as<c128>(h)->lv0->s, sfield);
        else st = get_field(as<double>(h), as<double>(h)->lv0->s, sfield);
    }
    emg3d_mg_destroy(h);
    return st;
}

int emg3d_mg_set_efield(emg3d_mg_t* mg, const void* e) {
    // one system: the whole field is overwritten, whatever layout it was in needs no conversion
    DISPATCH(mg, if (m->nsys == 1) m->lv0->e_home = 0; return set_field(m, m->sel_e(), e));
}
int emg3d_mg_get_efield(emg3d_mg_t* mg, void* e) { DISPATCH(mg, return get_field(m, m->sel_e(), e)); }

// ---- batched systems (several sources, one model): see common.hpp, Batch ---------------------------------
int emg3d_mg_set_batch(emg3d_mg_t* mg, int n) { DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->set_batch(n); }); }
int emg3d_mg_get_batch(emg3d_mg_t* mg) { DISPATCH(mg, return m->nsys); }
int emg3d_mg_select(emg3d_mg_t* mg, int b) { DISPATCH(mg, { if (b < 0 || b >= m->nsys) return -2; m->cur = b; return 0; }); }
int emg3d_mg_set_mask(emg3d_mg_t* mg, const int* active) {
    if (!active) return -2;
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->set_mask(active); });
}

int emg3d_mg_get_hfield(emg3d_mg_t* mg, int use_zeta, double smu0_re, double smu0_im, void* hfield) {
    if (!mg || !hfield || (smu0_re == 0.0 && smu0_im == 0.0)) return -2;
    DISPATCH(mg, {
        if (sizeof(*m->lv0->e) == 8 && smu0_im != 0.0) return -2;
        HIP_TRY(hipSetDevice(m->device));
        auto& L = *m->lv0;
        // the residual buffer is scratch between calls and large enough (nH < nE)
        launch_hfield(m->stream, L.nC, L.fl, m->sel_e(), use_zeta ? L.zeta : nullptr, L.h, L.ih, smu0_re, smu0_im, L.r);
        m->check_launch();
        HIP_TRY(m->d2h(hfield, L.r, (size_t)hfield_size(L.nC) * sizeof(*L.r)));
        return finish(m);
    });
}

int emg3d_interp3d(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* px, const double* py, const double* pz,
                   const void* values, int64_t n, const double* xi, int method, int has_fill, double fill_value,
                   double cval, void* out) {
    if (nx < 1 || ny < 1 || nz < 1 || n < 1 || !px || !py || !pz || !values || !xi || !out || method < 0 || method > 4) return -2;
    return dtype ? interp3d_host_impl<c128>(nx, ny, nz, px, py, pz, values, n, xi, method, has_fill, fill_value, cval, out)
                 : interp3d_host_impl<double>(nx, ny, nz, px, py, pz, values, n, xi, method, has_fill, fill_value, cval, out);
}

int emg3d_get_receiver_response(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx, const double* hy,
                                const double* hz, const double* origin, const void* field, int is_electric, int64_t n,
                                const double* xyz, const double* factors, void* resp) {
    if (nx < 3 || ny < 3 || nz < 3 || n < 1 || !hx || !hy || !hz || !field || !xyz || !factors || !resp) return -2;
    return dtype ? receiver_host_impl<c128>(nx, ny, nz, hx, hy, hz, origin, field, is_electric, n, xyz, factors, resp)
                 : receiver_host_impl<double>(nx, ny, nz, hx, hy, hz, origin, field, is_electric, n, xyz, factors, resp);
}

int emg3d_mg_get_receiver_response(emg3d_mg_t* mg, int magnetic, int use_zeta, double smu0_re, double smu0_im,
                                   int64_t n, const double* xyz, const double* factors, void* resp) {
    if (!mg || n < 1 || !xyz || !factors || !resp) return -2;
    if (magnetic && smu0_re == 0.0 && smu0_im == 0.0) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        auto& L = *m->lv0;
        if (L.nC[0] < 3 || L.nC[1] < 3 || L.nC[2] < 3) return -2;
        if (!m->scratch_field) m->scratch_field = m->template dalloc<T>(L.nE);
        const T* fdev = m->sel_e();
        if (magnetic) {        // H = get_h_field(E) into the residual buffer (scratch between calls, nH < nE)
            if (sizeof(T) == 8 && smu0_im != 0.0) return -2;
            launch_hfield(m->stream, L.nC, L.fl, m->sel_e(), use_zeta ? L.zeta : nullptr, L.h, L.ih, smu0_re, smu0_im, L.r);
            m->check_launch();
            fdev = L.r;
        }
        RcvComp<T> comp[3];
        receiver_components<T>(L.nodes, L.centers, L.nC, !magnetic, fdev, comp);
        const int rc = receiver_response_device<T>(m->stream, comp, n, xyz, factors, m->scratch_field, (T*)resp);
        const int st = finish(m);
        return rc ? rc : st;
    });
}

int emg3d_mg_get_residual(emg3d_mg_t* mg, void* r) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        m->residual(*m->lv0, 1, 0);
        int st = finish(m);
        if (st) return st;
        return get_field(m, m->sel_r(), r);
    });
}

// l2: one value per system of the batch (emg3d_mg_set_batch; 1 by default)
int emg3d_mg_residual_norm(emg3d_mg_t* mg, double* l2) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        m->residual(*m->lv0, 2, 0);
        return read_norms(m, m->nsys, l2);
    });
}

int emg3d_mg_sfield_norm(emg3d_mg_t* mg, double* l2) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        const int nb = 1024;
        if (nb > m->n_partials) { m->partials = m->template dalloc<double>(nb); m->n_partials = nb; }
        hipLaunchKernelGGL(k_abs2_partials<T>, dim3(nb), dim3(EMG_BLOCK), 0, m->stream, (const T*)m->sel_s(), m->lv0->nE, m->partials);
        hipLaunchKernelGGL(k_sum_sqrt, dim3(1), dim3(EMG_BLOCK), 0, m->stream, (const double*)m->partials, (i64)nb, m->norms, 0);
        m->check_launch();
        return read_norms(m, 1, l2);
    });
}

int emg3d_mg_smooth(emg3d_mg_t* mg, int nu, int lr_dir) {
    if (lr_dir < 0 || lr_dir > 7 || nu < 0) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        m->smoothing(*m->lv0, nu, lr_dir);
        return finish(m);
    });
}

int emg3d_mg_cycle(emg3d_mg_t* mg, int sc_dir, int lr_dir, double* l2) {
    if (sc_dir < 0 || sc_dir > 3 || lr_dir < 0 || lr_dir > 7) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        const bool tlog = getenv("EMG3D_LOG_SETUP") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        m->cycle0(sc_dir, lr_dir, 0);
        const auto t1 = std::chrono::steady_clock::now();
        const int st = read_norms(m, m->nsys, l2);
        if (tlog) fprintf(stderr, "[cycle] (%d,%d): enqueue %.2f ms, until the norm is back %.2f ms\n", sc_dir, lr_dir,
                          std::chrono::duration<double, std::milli>(t1 - t0).count(),
                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return st;
    });
}

int emg3d_mg_cycle_next(emg3d_mg_t* mg, int sc_dir, int lr_dir, int next_sc_dir, int next_lr_dir, double* l2) {
    if (sc_dir < 0 || sc_dir > 3 || lr_dir < 0 || lr_dir > 7 || next_sc_dir > 3 || next_lr_dir > 7 || !l2) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        const bool nxt = next_sc_dir >= 0 && next_lr_dir >= 0;
        m->cycle_then_prepare(sc_dir, lr_dir, nxt ? next_sc_dir : -1, nxt ? next_lr_dir : -1, l2);
        const int e = m->err;          // (the stream is NOT synchronised here: the next pair's factor kernels may still run)
        m->err = 0;
        return e;
    });
}

int emg3d_mg_set_trace(emg3d_mg_t* mg, int on) {
    DISPATCH(mg, { m->trace = on != 0; m->trace_recs.clear(); m->trace_dropped = 0; return 0; });
}

int emg3d_mg_get_trace(emg3d_mg_t* mg, int max_recs, int64_t* recs, double* norms, int* count) {
    if (max_recs < 0 || !recs || !norms || !count) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        const int n = std::min<int>((int)m->trace_recs.size(), max_recs);
        if (n > 0) HIP_TRY(hipMemcpyAsync(norms, m->trace_norms, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        const int st = finish(m);
        for (int i = 0; i < n; ++i) {
            const auto& r = m->trace_recs[(size_t)i];
            int64_t* o = recs + 7 * (size_t)i;
            o[0] = r.it; o[1] = r.level; o[2] = r.cycmax; o[3] = r.kind; o[4] = r.n[0]; o[5] = r.n[1]; o[6] = r.n[2];
        }
        *count = n;
        // records that did not fit (the caller's buffer or the handle's TRACE_MAX): reported, not dropped silently
        if (m->trace_dropped > 0 || (int)m->trace_recs.size() > n)
            fprintf(stderr, "[emg3d_hip] trace: %d record(s) lost (buffer of %d, %d recorded, %d beyond the handle's limit)\n",
                    (int)m->trace_recs.size() - n + m->trace_dropped, max_recs, (int)m->trace_recs.size(), m->trace_dropped);
        m->trace_recs.clear();
        m->trace_dropped = 0;
        return st;
    });
}

int emg3d_mg_prepare(emg3d_mg_t* mg, int sc_dir, int lr_dir) {
    if (sc_dir < 0 || sc_dir > 3 || lr_dir < 0 || lr_dir > 7) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        m->prepare(sc_dir, lr_dir);
        return finish(m);
    });
}

int emg3d_mg_cycles(emg3d_mg_t* mg, int ncycles, const int* sc_cycle, int n_sc, const int* lr_cycle, int n_lr,
                    double* l2) {
    if (ncycles < 1 || ncycles > 4095 || n_sc < 1 || n_lr < 1) return -2;
    DISPATCH(mg, {
        if ((ncycles + 1) * m->nsys > MG<T>::NORM_SLOTS) return -2;
        HIP_TRY(hipSetDevice(m->device));
        const auto t0 = std::chrono::steady_clock::now();
        // every replayed graph writes its norm to slot 0 and is copied from there: cycle i keeps slot i + 1
        for (int i = 0; i < ncycles; ++i) m->cycle0(sc_cycle[i % n_sc], lr_cycle[i % n_lr], i + 1);
        const auto t1 = std::chrono::steady_clock::now();
        HIP_TRY(hipMemcpyAsync(l2, m->norms + m->nsys, (size_t)ncycles * m->nsys * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        const int st = finish(m);
        if (m->log_launches) {
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[cycles] %d cycles: host enqueue %.3f ms, until results %.3f ms\n", ncycles,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t0).count());
        }
        return st;
    });
}

void* emg3d_mg_efield_devptr(emg3d_mg_t* mg) {
    if (!mg) return nullptr;
    return reinterpret_cast<emg3d_mg*>(mg)->dtype ? efield_ptr(as<c128>(mg)) : efield_ptr(as<double>(mg));
}
void* emg3d_mg_sfield_devptr(emg3d_mg_t* mg) {
    if (!mg) return nullptr;
    return reinterpret_cast<emg3d_mg*>(mg)->dtype ? (void*)as<c128>(mg)->sel_s() : (void*)as<double>(mg)->sel_s();
}
void* emg3d_mg_stream(emg3d_mg_t* mg) {
    if (!mg) return nullptr;
    return reinterpret_cast<emg3d_mg*>(mg)->dtype ? (void*)as<c128>(mg)->stream : (void*)as<double>(mg)->stream;
}
int64_t emg3d_mg_nE(emg3d_mg_t* mg) { DISPATCH(mg, return m->lv0->nE); }
int emg3d_mg_sync(emg3d_mg_t* mg) { DISPATCH(mg, return finish(m)); }
int64_t emg3d_mg_device_bytes(emg3d_mg_t* mg) { DISPATCH(mg, return m->bytes); }

int emg3d_mg_time_sweep(emg3d_mg_t* mg, int dir, int reps, float* ms_per_sweep) {
    if (dir < 0 || dir > 3 || reps < 1) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        // warm-up (also builds the cached factorisation outside the timed region);
        // the conversion to/from the working copy (x-lines) is outside the timed
        // region too: the events bracket launches of the sweep kernel only.
        if (dir == 0) m->smooth_point(L, 1); else m->smooth_line(L, dir - 1, 1);
        if (dir > 0) m->to_work(L, dir - 1);
        hipEvent_t t0; hipEvent_t t1;
        HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
        HIP_TRY(hipEventRecord(t0, m->stream));
        for (int i = 0; i < reps; ++i) { if (dir == 0) m->smooth_point(L, 1); else m->smooth_line(L, dir - 1, 1, false, false); }
        HIP_TRY(hipEventRecord(t1, m->stream));
        HIP_TRY(hipEventSynchronize(t1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
        hipEventDestroy(t0); hipEventDestroy(t1);
        *ms_per_sweep = ms / reps;
        if (dir > 0) m->from_work(L, dir - 1);
        return finish(m);
    });
}

int emg3d_mg_placement(emg3d_mg_t* mg, int w, int* tries, int* kept, float* ms) {
    if (w < 0 || w > 1 || !tries || !kept || !ms) return -2;
    DISPATCH(mg, {
        const auto& R = m->place_rec[w];
        *tries = R.tries; *kept = R.reused ? -1 : R.kept;
        for (int k = 0; k < MG<T>::PLACE_MAX; ++k) ms[k] = k < R.tries ? R.ms[k] : 0.f;
        return 0;
    });
}

int emg3d_mg_last_sweep_kernel(emg3d_mg_t* mg, char* name) {
    if (!name) return -2;
    DISPATCH(mg, { strncpy(name, m->sweep_name, 63); name[63] = 0; return 0; });
}

int emg3d_mg_last_residual_kernel(emg3d_mg_t* mg, char* name) {
    if (!name) return -2;
    DISPATCH(mg, { strncpy(name, m->res_name, 63); name[63] = 0; return 0; });
}

int emg3d_mg_time_residual(emg3d_mg_t* mg, int reps, float* ms_per_call) {
    if (reps < 1) return -2;
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        m->residual(L, 1, 0);
        hipEvent_t t0; hipEvent_t t1;
        HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
        HIP_TRY(hipEventRecord(t0, m->stream));
        for (int i = 0; i < reps; ++i) m->residual(L, 1, 0);
        HIP_TRY(hipEventRecord(t1, m->stream));
        HIP_TRY(hipEventSynchronize(t1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
        hipEventDestroy(t0); hipEventDestroy(t1);
        *ms_per_call = ms / reps;
        return finish(m);
    });
}

int emg3d_mg_amatvec(emg3d_mg_t* mg, const void* x_host, void* y_host) {
    DISPATCH(mg, {
        // y = -(0 - A x) = A x : residual with s = 0 (solver.py:646-660)
        HIP_TRY(hipSetDevice(m->device));
        Level<T>& L = *m->lv0;
        // use r as output, a temporary zero source: compute r = 0 - A x via MODE 0 on zeroed r
        HIP_TRY(hipMemsetAsync(L.r, 0, (size_t)L.nE * sizeof(T), m->stream));
        if (!m->scratch_field) m->scratch_field = m->template dalloc<T>(L.nE);
        T* x = m->scratch_field;
        HIP_TRY(m->h2d(x, x_host, (size_t)L.nE * sizeof(T)));
        ResidualArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = L.nC[q]; a.eta[q] = L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q]; }
        a.fl = L.fl; a.r = L.r; a.s = L.r; a.e = x; a.zeta = L.zeta; a.partials = nullptr;
        const i64 plane = (L.nC[0] + 1) * (L.nC[1] + 1);
        dim3 grid((unsigned)((plane + EMG_BLOCK - 1) / EMG_BLOCK), (unsigned)(L.nC[2] + 1));
        residual_launch<T>(0, 1, grid, m->stream, a);
        hipLaunchKernelGGL(k_negate<T>, dim3(1024), dim3(EMG_BLOCK), 0, m->stream, L.r, L.nE);
        m->check_launch();
        int st = finish(m);
        if (st) return st;
        return get_field(m, L.r, y_host);
    });
}

// ---- Krylov vector workspace (device-resident BiCGSTAB) ----------------

int emg3d_mg_vec_alloc(emg3d_mg_t* mg, int n) {
    if (n < 0 || n > 256) return -2;        // gcrotmk(m = 20, k = 20): 5 + 41 + 40 + 42 vectors at most
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_alloc(n); });
}
int emg3d_mg_vec_set(emg3d_mg_t* mg, int id, const void* host) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        T* v = m->vec(id);
        if (!v || !host) return -2;
        HIP_TRY(m->h2d(v, host, (size_t)m->lv0->nE * sizeof(T)));
        m->touched(id);
        return finish(m);
    });
}
int emg3d_mg_vec_get(emg3d_mg_t* mg, int id, void* host) {
    DISPATCH(mg, {
        HIP_TRY(hipSetDevice(m->device));
        T* v = m->vec(id);
        if (!v || !host) return -2;
        return get_field(m, v, host);
    });
}
int emg3d_mg_vec_copy(emg3d_mg_t* mg, int dst, int src) {
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_copy(dst, src); });
}
int emg3d_mg_vec_axpy(emg3d_mg_t* mg, int y, double alpha_re, double alpha_im, int x) {
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_axpy(y, scalar_of<T>(alpha_re, alpha_im), x); });
}
int emg3d_mg_vec_scale(emg3d_mg_t* mg, int y, double alpha_re, double alpha_im) {
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_scale(y, scalar_of<T>(alpha_re, alpha_im)); });
}
int emg3d_mg_vec_dot(emg3d_mg_t* mg, int a, int b, double* out2) {
    if (!out2) return -2;
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_dot(a, b, out2); });
}
int emg3d_mg_vec_amatvec(emg3d_mg_t* mg, int dst, int src) {
    DISPATCH(mg, { HIP_TRY(hipSetDevice(m->device)); return m->vec_amatvec(dst, src); });
}

}  // extern "C"
