// Common device/host types for the emg3d MI355X (gfx950) hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include <cstdlib>
#include <type_traits>

typedef int64_t i64;

// Tuning knobs exist only in the lab build (-DEMG3D_LAB, libemg3d_hip_lab.so): there every LAB_ENV reads its environment
// variable; the product library takes the default and does not even contain the variable's name.
#ifdef EMG3D_LAB
inline long long lab_env_(const char* name, long long def) { const char* v = getenv(name); return v ? atoll(v) : def; }
inline int lab_env_ch_(const char* name) { const char* v = getenv(name); return v ? v[0] : 0; }
#define LAB_ENV(name, def) lab_env_(name, (long long)(def))
#define LAB_ENV_CH(name) lab_env_ch_(name)
#else
#define LAB_ENV(name, def) ((long long)(def))
#define LAB_ENV_CH(name) 0
#endif

#define HD __host__ __device__ __forceinline__

// complex128 as two doubles (interleaved, layout-compatible with numpy).
struct c128 {
    double re, im;
};

HD c128 mk(double re, double im) { c128 r; r.re = re; r.im = im; return r; }
HD c128 operator+(c128 a, c128 b) { return mk(a.re + b.re, a.im + b.im); }
HD c128 operator-(c128 a, c128 b) { return mk(a.re - b.re, a.im - b.im); }
HD c128 operator-(c128 a) { return mk(-a.re, -a.im); }
HD c128 operator*(c128 a, c128 b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
HD c128 operator*(double a, c128 b) { return mk(a * b.re, a * b.im); }
HD c128 operator*(c128 a, double b) { return mk(a.re * b, a.im * b); }
HD c128 operator/(c128 a, double b) { return mk(a.re / b, a.im / b); }
HD c128& operator+=(c128& a, c128 b) { a.re += b.re; a.im += b.im; return a; }
HD c128& operator-=(c128& a, c128 b) { a.re -= b.re; a.im -= b.im; return a; }
HD c128& operator*=(c128& a, c128 b) { a = a * b; return a; }
HD c128& operator*=(c128& a, double b) { a.re *= b; a.im *= b; return a; }

// acc += a * b with fused multiply-adds (4 FMAs for complex)
HD void cmac(double& acc, double a, double b) { acc = __builtin_fma(a, b, acc); }
HD void cmac(c128& acc, c128 a, c128 b) {
    acc.re = __builtin_fma(-a.im, b.im, __builtin_fma(a.re, b.re, acc.re));
    acc.im = __builtin_fma(a.im, b.re, __builtin_fma(a.re, b.im, acc.im));
}
HD void cmac(c128& acc, c128 a, double b) {
    acc.re = __builtin_fma(a.re, b, acc.re);
    acc.im = __builtin_fma(a.im, b, acc.im);
}
// acc -= a * b
HD void cmsc(double& acc, double a, double b) { acc = __builtin_fma(-a, b, acc); }
HD void cmsc(c128& acc, c128 a, c128 b) {
    acc.re = __builtin_fma(a.im, b.im, __builtin_fma(-a.re, b.re, acc.re));
    acc.im = __builtin_fma(-a.im, b.re, __builtin_fma(-a.re, b.im, acc.im));
}
HD void cmsc(c128& acc, c128 a, double b) {
    acc.re = __builtin_fma(-a.re, b, acc.re);
    acc.im = __builtin_fma(-a.im, b, acc.im);
}

HD double recip(double a) { return 1.0 / a; }
HD c128 recip(c128 a) {
    const double d = 1.0 / (a.re * a.re + a.im * a.im);
    return mk(a.re * d, -a.im * d);
}
HD double abs2(double a) { return a * a; }
HD double abs2(c128 a) { return a.re * a.re + a.im * a.im; }

typedef double emg_d2 __attribute__((ext_vector_type(2)));

template <class T> struct Zero;
template <> struct Zero<double> { HD static double v() { return 0.0; } };
template <> struct Zero<c128> { HD static c128 v() { return mk(0.0, 0.0); } };
HD double real_of(double a) { return a; }
HD double real_of(c128 a) { return a.re; }
HD void add_real(double& a, double r) { a += r; }
HD void add_real(c128& a, double r) { a.re += r; }

// Where the three field components and the cell arrays live.  Strides are
// explicit so that the same kernels run on the reference layout (x fastest)
// and on axis-permuted working copies.
struct FieldLayout {
    i64 off[3];
    i64 st[3][3];
};
struct CellLayout {
    i64 st[3];
};

struct GridDims {
    i64 nC[3];
    i64 nN[3];
};

inline FieldLayout ref_field_layout(const i64 nC[3]) {
    FieldLayout f;
    i64 o = 0;
    for (int c = 0; c < 3; ++c) {
        i64 d[3];
        for (int a = 0; a < 3; ++a) d[a] = (a == c) ? nC[a] : nC[a] + 1;
        f.off[c] = o;
        f.st[c][0] = 1; f.st[c][1] = d[0]; f.st[c][2] = d[0] * d[1];
        o += d[0] * d[1] * d[2];
    }
    return f;
}
inline CellLayout ref_cell_layout(const i64 nC[3]) {
    CellLayout c;
    c.st[0] = 1; c.st[1] = nC[0]; c.st[2] = nC[0] * nC[1];
    return c;
}
inline i64 n_edges(const i64 nC[3]) {
    return nC[0] * (nC[1] + 1) * (nC[2] + 1) + (nC[0] + 1) * nC[1] * (nC[2] + 1) +
           (nC[0] + 1) * (nC[1] + 1) * nC[2];
}

// lin -> (i0, i1, i2) of an n0 x n1 x n2 box, i0 fastest.  A 64-bit division by a runtime value is a ~100-instruction sequence,
// three of them were half of the restriction / prolongation kernels' instruction streams (launches of 5-8 us on the coarse levels):
// 32-bit divisions whenever the box has fewer than 2^32 points (a wave-uniform test).
__device__ __forceinline__ void unlin3(i64 lin, i64 n0, i64 n1, i64 n2, i64& i0, i64& i1, i64& i2) {
    if (n0 * n1 * n2 <= (i64)0xffffffffll) {
        const unsigned l = (unsigned)lin, m0 = (unsigned)n0, m1 = (unsigned)n1;
        const unsigned q = l / m0, q2 = q / m1;
        i0 = (i64)(l - q * m0); i1 = (i64)(q - q2 * m1); i2 = (i64)q2;
    } else {
        i0 = lin % n0; i1 = (lin / n0) % n1; i2 = lin / (n0 * n1);
    }
}
// lin -> (i0, i1) with i0 fastest
__device__ __forceinline__ void unlin2(i64 lin, i64 n0, i64& i0, i64& i1) {
    if (lin <= (i64)0xffffffffll && n0 <= (i64)0xffffffffll) {
        const unsigned l = (unsigned)lin, m0 = (unsigned)n0, q = l / m0;
        i0 = (i64)(l - q * m0); i1 = (i64)q;
    } else { i1 = lin / n0; i0 = lin - i1 * n0; }
}

// Parity split of an index range [0, n): even indices first, then the odd ones.
// Same-colour lines (index step 2) become contiguous in memory.
HD i64 psplit(i64 v, i64 n) { return (v & 1) ? ((n + 1) >> 1) + (v >> 1) : (v >> 1); }

// Batched systems: several right-hand sides (sources) that share the grid, the model and therefore the cached line
// factorisations run through the SAME launches -- every field array of a level is [system][nE], the kernels of the
// cycle take the system index from a grid dimension.  A 7 us coarse-level launch then does nb x the work, and the
// workgroups of the nb systems that sweep the same lines sit on the same XCD (the grids are multiples of 8), so the
// factor is fetched from HBM once and served to the other systems by the L2.  st == 0: single system (no index).
// mask[b] == 0: system b is frozen (it converged earlier; its field must stay bit for bit what a solve of its own
// would return).
struct Batch {
    i64 st = 0;                 // elements between consecutive systems of this level's field arrays (0: one system)
    const int* mask = nullptr;  // device, one flag per system, or nullptr
    int n = 1;                  // systems
};
// System index from grid dimension DIM (x|y|z): defines b_ and boff_ (element offset of the system's arrays).
#define EMG_BATCH(DIM, bt)                                              \
    const int b_ = (bt).st ? (int)blockIdx.DIM : 0;                     \
    if ((bt).mask && !(bt).mask[b_]) return;                            \
    const i64 boff_ = (i64)b_ * (bt).st;
// The line-sweep kernels fold the system into blockIdx.x instead (a launch has G * n workgroups): workgroup bx runs on
// XCD bx % 8; within an XCD the n systems of one line block follow each other, so that the block's factor entries
// are still in that XCD's L2 (4 MB) when the next system asks for them -- with the system in blockIdx.y a whole
// colour of the factor (24 MB at 128^3) would pass between two uses.  Defines wg (line block), boff_ (the system's offset in the field arrays) and bsys_ (the system).
#define EMG_SWEEP_WG(a)                                                                                     \
    i64 wg;                                                                                                 \
    i64 boff_ = 0;                                                                                          \
    unsigned bsys_ = 0;                                                                                     \
    if ((a).bt.st) {                                                                                        \
        const unsigned n_ = (unsigned)(a).bt.n, G_ = gridDim.x / n_;                                        \
        const unsigned seq_ = (a).xcd ? blockIdx.x >> 3 : blockIdx.x;                                       \
        const unsigned j_ = seq_ / n_;                                                                      \
        const unsigned b_ = seq_ - j_ * n_;                                                                 \
        wg = (a).xcd ? (i64)(blockIdx.x & 7) * ((G_ + 7) >> 3) + j_ : (i64)j_;                              \
        if ((a).bt.mask && !(a).bt.mask[b_]) return;                                                        \
        boff_ = (i64)b_ * (a).bt.st;                                                                        \
        bsys_ = b_;                                                                                         \
    } else {                                                                                                \
        wg = (a).xcd ? (i64)(blockIdx.x & 7) * ((gridDim.x + 7) >> 3) + (blockIdx.x >> 3) : (i64)blockIdx.x; \
    }
// (the kernels add boff_ where they read a.e / a.s: writing to a member of the argument struct would move the whole
// struct from the kernarg segment to scratch memory)

// Kernel arguments in ONE burst.  The argument structs are passed by value (200-600 bytes of kernel-argument segment); the
// compiler loads a member where it is first used, and every early exit or mode branch in front of the arithmetic waits for "its"
// member before the next one is requested: 6-10 dependent scalar round trips in the prologue of kernels that live for 4-8 us on
// the coarse levels (420 + ~100 such launches per 128^3 F-cycle; k_line_sweep_qpl: 6.4 -> 5.3 us per launch, the cycle 10.34 ->
// 9.6 ms).  Naming the members as SGPR operands of an empty asm makes all scalar loads issue back to back at the top with one
// wait (an asm takes at most 30 operands).
#define EMG_ARGS_BURST(...) asm volatile("" :: __VA_ARGS__)
#define EMG_S(x) "s"(x)

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            fprintf(stderr, "[emg3d_hip] %s failed at %s:%d: %s\n", #expr, __FILE__,    \
                    __LINE__, hipGetErrorString(_e));                                   \
            (void)hipGetLastError();    /* reported: not left for the runtime's next user */ \
            return (int)_e;                                                             \
        }                                                                               \
    } while (0)

// Device memory of the stateless entry points: freed on every return path (HIP_TRY returns early).
struct DevBlock {
    void* p = nullptr;
    DevBlock() = default;
    DevBlock(const DevBlock&) = delete;
    DevBlock& operator=(const DevBlock&) = delete;
    ~DevBlock() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t nb) { return hipMalloc(&p, nb ? nb : 1); }
    template <class U> U* get() const { return (U*)p; }
};
// ptr = a device block of `bytes` that is freed when the enclosing scope is left (ptr must be a declared pointer)
#define DEV_ALLOC(ptr, bytes)               \
    DevBlock ptr##_blk;                     \
    HIP_TRY(ptr##_blk.alloc(bytes));        \
    ptr = ptr##_blk.get<typename std::remove_pointer<decltype(ptr)>::type>()
