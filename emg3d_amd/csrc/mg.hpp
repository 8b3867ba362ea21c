// Device-resident multigrid hierarchy and cycle driver.
//
// Restates the control flow of solver.multigrid / smoothing / restriction /
// prolongation / residual (reference emg3d/solver.py:434-607, 738-1039) with
// every array resident in HBM: the host only enqueues kernels on one stream.
// The reference rebuilds the coarse grid, model, weights and prolongator at
// every visit (solver.py:859-899, 933-963); here each hierarchy (one per
// global semicoarsening direction var.sc_dir in 0..3) is built once, lazily,
// and the line factorisations are cached per (level, direction).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <chrono>
#include <vector>

#include "common.hpp"
#include "smooth.hpp"
#include "stencil.hpp"
#include "smooth_qpl.hpp"
#include "smooth_qc.hpp"
#include "smooth_thm.hpp"
#include "smooth_tha.hpp"
#include "sweep_launch.hpp"     // the four sweep families are compiled in units of their own; this file launches through these

template <class T>
struct Level {
    i64 nC[3];
    i64 nE, nCells;
    FieldLayout fl;
    CellLayout cl;
    std::vector<double> h_host[3], nodes[3], centers[3];
    double* h[3] = {nullptr, nullptr, nullptr};
    double* ih[3] = {nullptr, nullptr, nullptr};   // 1/h
    T* eta[3] = {nullptr, nullptr, nullptr};
    double* zeta = nullptr;
    unsigned char* sflag[3] = {nullptr, nullptr, nullptr};   // level 0: per line direction, which lines carry a source
    bool sflag_valid[3] = {false, false, false};
    bool s_dense = false;       // the source came from a Krylov vector (emg3d_mg_vec_copy): every line counts as carrying one
    bool zeta_sep = false;      // zeta == (hx hy) hz bit for bit (level 0 of a model without mu_r): MG::check_zeta
    T *s = nullptr, *e = nullptr, *r = nullptr;
    // x<->y transposed working copies (y fastest) for line relaxation along x:
    // lanes run across lines, so the transverse axis must be the contiguous one.
    T *eT = nullptr, *sT = nullptr;
    T* etaT[3] = {nullptr, nullptr, nullptr};
    double* zetaT = nullptr;
    bool sT_valid = false;
    FieldLayout flT;
    CellLayout clT;
    // parity-split working copies for the sweeps: [0] x-lines: (y-split, x, z);
    // [1] y-/z-lines: (x-split, y, z).  Same strides as flT / fl.
    T *eW[2] = {nullptr, nullptr}, *sW[2] = {nullptr, nullptr};
    double* zetaW[2] = {nullptr, nullptr};
    bool sW_valid[2] = {false, false};
    // levels with split working copies: where the field currently lives -- 0: e (reference layout), 1: eW[1] (x-split;
    // e is then stale).  Level 0 moves between the two (MG::home_on); the coarse levels are at home in eW[1] / sW[1] for
    // good (MG::home_lvl: the restriction writes there, the prolongation reads there; e and s are never touched).
    int e_home = 0;
    // cached line factorisations
    T* fac[3] = {nullptr, nullptr, nullptr};
    // per-lane launch descriptors of the scan kernel's colour launches (LineArgs::qd; smooth_qpl.hpp DM): [direction][colour]
    void* qd[3][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    unsigned qdn[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    i64 fac_lines[3] = {0, 0, 0};
    i64 fac_mid[3] = {0, 0, 0};   // middle block of the (two-sided) factorisation
    int fac_kind[3] = {0, 0, 0};  // 0: one-sided, 15 numbers per block (k_line_sweep_rp / _qpl, k_line_sweep); 3: mirrored
                                  // two-sided (k_line_factor_m, for k_line_sweep_thm); 4: one-sided compact (11 numbers per
                                  // block, k_line_sweep_qc)
};

// Transfer operators between a level and the next coarser one of a hierarchy.
struct Transfer {
    int sc = -1;         // current sc_dir (0..6) used to build the child
    int co[3] = {0, 0, 0};
    double* w[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int* pidx[3] = {nullptr, nullptr, nullptr};
    double* pwt[3] = {nullptr, nullptr, nullptr};
};

template <class T>
struct Hierarchy {
    std::vector<std::shared_ptr<Level<T>>> lv;
    std::vector<Transfer> tr;     // tr[l]: between lv[l] and lv[l+1]
};

// Freed device blocks are kept for the next handle of the process (exact-size reuse).  Solves come in
// series on the same grid (frequencies, sources, Krylov restarts), so a new handle asks for exactly the
// sizes the last one returned; going through hipFree / hipMalloc instead costs ~13 ms per close of a 128^3
// handle plus the allocation time of the next one.  Bounded by EMG3D_POOL_GB (default 96 GiB per process; 0 disables);
// emg3d_hip_release_cached() returns everything to the driver.
class DevicePool {
    std::mutex mu;
    // A parked block remembers the ROLE it was placed for (MG::place_level0: tag 1 / 2 = the working copy the x- / the y- and z-line
    // sweeps of a large level 0 write; 0 = none): the next handle of the same size asks for its working copies by role and gets the
    // blocks the last handle had searched for, instead of searching again.
    struct Parked { void* p; int tag; };
    std::multimap<std::pair<int, size_t>, Parked> free_blocks;     // device < 0: pinned host memory of device -1 - key
    size_t held = 0, cap;
public:
    DevicePool() {
        const char* g = getenv("EMG3D_POOL_GB");
        cap = (size_t)((g ? atof(g) : 96.0) * (double)((size_t)1 << 30));
    }
    // tag != 0: a block parked with that tag if there is one (*hit = true), else an untagged one, else any; tag == 0: untagged
    // blocks first (the tagged ones stay for the working copies that will ask for them)
    void* take(int device, size_t nb, int tag = 0, bool* hit = nullptr) {
        std::lock_guard<std::mutex> lk(mu);
        if (hit) *hit = false;
        auto rng = free_blocks.equal_range({device, nb});
        if (rng.first == rng.second) return nullptr;
        auto pick = rng.second;
        for (auto it = rng.first; it != rng.second; ++it) if (it->second.tag == tag) { pick = it; break; }
        if (pick == rng.second && tag != 0)
            for (auto it = rng.first; it != rng.second; ++it) if (it->second.tag == 0) { pick = it; break; }
        if (pick == rng.second) pick = rng.first;
        if (hit) *hit = tag != 0 && pick->second.tag == tag;
        void* p = pick->second.p;
        free_blocks.erase(pick);
        held -= nb;
        return p;
    }
    bool give(int device, void* p, size_t nb, int tag = 0) {     // false: not kept, the caller frees
        std::lock_guard<std::mutex> lk(mu);
        if (held + nb > cap) return false;
        free_blocks.insert({{device, nb}, Parked{p, tag}});
        held += nb;
        return true;
    }
    size_t release_all() {
        std::lock_guard<std::mutex> lk(mu);
        const size_t n = held;
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& kv : free_blocks) {
            if (kv.first.first < 0) { (void)hipHostFree(kv.second.p); continue; }
            (void)hipSetDevice(kv.first.first); (void)hipFree(kv.second.p);
        }
        for (auto& kv : free_streams) { (void)hipSetDevice(kv.first); (void)hipStreamDestroy(kv.second); }
        free_streams.clear();
        (void)hipSetDevice(cur);
        free_blocks.clear();
        held = 0;
        return n;
    }
    size_t bytes_held() { std::lock_guard<std::mutex> lk(mu); return held; }
    size_t bytes_held_on(int device) {      // device memory parked for `device` (keys < 0 are pinned host blocks)
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        for (auto& kv : free_blocks) if (kv.first.first == device) n += kv.first.second;
        return n;
    }
    // HIP streams of closed handles (creating and destroying one costs about a millisecond each; a solve makes two):
    // idle -- their handle synchronised them before giving them back -- and non-blocking
    std::multimap<int, hipStream_t> free_streams;
    hipStream_t take_stream(int device) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_streams.find(device);
            if (it != free_streams.end()) { hipStream_t s = it->second; free_streams.erase(it); return s; }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return s;
    }
    void give_stream(int device, hipStream_t s) {
        std::lock_guard<std::mutex> lk(mu);
        if (free_streams.count(device) < 16) free_streams.insert({device, s});
        else (void)hipStreamDestroy(s);
    }
    static DevicePool& get() { static DevicePool* p = new DevicePool(); return *p; }   // never destroyed: no HIP calls at exit
};

inline int current_sc_dir(int sc_dir, const i64 nC[3]) {   // solver.py:1467-1514
    const bool xs = nC[0] % 2 != 0 || nC[0] < 3 || sc_dir == 1;
    const bool ys = nC[1] % 2 != 0 || nC[1] < 3 || sc_dir == 2;
    const bool zs = nC[2] % 2 != 0 || nC[2] < 3 || sc_dir == 3;
    if (xs) { if (ys) return 6; return zs ? 5 : 1; }
    if (ys) return zs ? 4 : 2;
    return zs ? 3 : 0;
}

inline int current_lr_dir(int lr, const i64 nC[3]) {       // solver.py:1517-1572
    if (nC[0] == 2) { if (lr == 1) lr = 0; else if (lr == 5) lr = 3; else if (lr == 6) lr = 2; else if (lr == 7) lr = 4; }
    if (nC[1] == 2) { if (lr == 2) lr = 0; else if (lr == 4) lr = 3; else if (lr == 6) lr = 1; else if (lr == 7) lr = 5; }
    if (nC[2] == 2) { if (lr == 3) lr = 0; else if (lr == 4) lr = 2; else if (lr == 5) lr = 1; else if (lr == 7) lr = 6; }
    return lr;
}

inline void sc_axes(int sc, int co[3]) {                    // solver.py:850-856
    co[0] = !(sc == 1 || sc == 5 || sc == 6);
    co[1] = !(sc == 2 || sc == 4 || sc == 6);
    co[2] = !(sc == 3 || sc == 4 || sc == 5);
}

// core.restrict_weights, emg3d/core.py:1970-2041 (O(n) host work).
inline void restrict_weights_host(const double* vectorN, const double* vectorCC, const double* h, i64 nh,
                                  const double* cvectorN, const double* cvectorCC, const double* ch,
                                  i64 n, double* wl, double* w0, double* wr) {
    std::vector<double> d(n + 1);
    d[0] = h[0] / 2; d[n] = h[nh - 1] / 2;
    for (i64 i = 1; i < n; ++i) d[i] = (h[2 * i - 2] + h[2 * i - 1]) / 2.;
    for (i64 i = 0; i < n; ++i) wl[i] = 1 / d[i];
    wl[0] *= (vectorN[0] - h[0] / 2) - (cvectorN[0] - ch[0] / 2);
    for (i64 i = 1; i < n; ++i) wl[i] *= vectorCC[2 * i - 1] - cvectorCC[i - 1];
    for (i64 i = 0; i < n; ++i) w0[i] = 1.0;
    for (i64 i = 0; i < n; ++i) wr[i] = 1 / d[i + 1];
    wr[n - 1] *= (cvectorN[n - 1] + ch[n - 2] / 2) - (vectorN[nh] + h[nh - 1] / 2);
    for (i64 i = 0; i < n - 1; ++i) wr[i] *= cvectorCC[i] - vectorCC[2 * i];
}

// RegularGridProlongator._set_edges_and_weights, solver.py:1432-1447:
// i = searchsorted(c, x) - 1 clipped to [0, size-2]; t = (x - c[i])/(c[i+1]-c[i]).
inline void prolong_weights_host(const std::vector<double>& cn, const std::vector<double>& fn,
                                 std::vector<int>& idx, std::vector<double>& wt) {
    idx.resize(fn.size()); wt.resize(fn.size());
    for (size_t j = 0; j < fn.size(); ++j) {
        i64 i = (i64)(std::lower_bound(cn.begin(), cn.end(), fn[j]) - cn.begin()) - 1;
        if (i < 0) i = 0;
        if (i > (i64)cn.size() - 2) i = (i64)cn.size() - 2;
        idx[j] = (int)i;
        wt[j] = (fn[j] - cn[i]) / (cn[i + 1] - cn[i]);
    }
}

struct emg3d_mg {
    virtual ~emg3d_mg() {}
    int dtype = 1;
};

// Every kernel launch of a handle goes through MG_LAUNCH: after a failed device allocation (MG::broken) some array of the
// handle is missing, and a kernel that touches a null pointer takes the whole PROCESS down with a memory fault instead of
// returning the error.  A broken handle launches nothing; every entry point answers hipErrorOutOfMemory until it is destroyed.
#define MG_LAUNCH(...) do { if (!broken) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

template <class T>
struct MG : emg3d_mg {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double origin[3] = {0, 0, 0};
    std::shared_ptr<Level<T>> lv0;
    std::map<int, Hierarchy<T>> hier;   // key: global sc_dir
    std::vector<std::pair<void*, size_t>> allocs;
    i64 bytes = 0;
    // parameters (MGParameters subset)
    int cycle = 'F', cycmax = 2, nu_init = 0, nu_pre = 2, nu_coarse = 1, nu_post = 2, order = 1;
    int clevel[4] = {0, 0, 0, 0};
    bool eta_alias[3] = {false, false, false};   // eta_y/eta_z alias eta_x
    // scratch
    double* partials = nullptr; i64 n_partials = 0;
    double* norms = nullptr;    // device
    T* scratch_field = nullptr; // nE scratch (Krylov matvec input)
    static const int NORM_SLOTS = 4096;
    int err = 0;
    bool broken = false;        // a device allocation failed: arrays are missing, nothing is launched any more (MG_LAUNCH)
    // ---- kernel selection -----------------------------------------------------------------------------------------
    // The product library runs the measured defaults (why each is what it is: DESIGN.md 3; the A/B numbers behind them:
    // profiles/HISTORY.md) and reads six documented variables: EMG3D_POOL_GB, EMG3D_GRAPH, EMG3D_LOG, EMG3D_LOG_SETUP,
    // EMG3D_BATCH_TUNE, EMG3D_PLACE_TRIES.  The lab build (-DEMG3D_LAB: libemg3d_hip_lab.so, used by tests/test_gpu_variants.py and
    // tools/) also compiles the superseded kernels and reads one variable per knob below (LAB_ENV, common.hpp).
    int sweep_kernel = LAB_ENV_CH("EMG3D_SWEEP") == 't' ? 1 : 0;       // 1: thread-per-line kernel everywhere
    bool use_xt = LAB_ENV("EMG3D_XT", 1) != 0;                          // x-lines on x<->y transposed working copies ...
    i64 xt_min_cells = LAB_ENV("EMG3D_XT_MIN", 8192);                   // ... on levels of at least this many cells
    int th_lpw = (int)LAB_ENV("EMG3D_TH_LPW", 0);                       // lines per pair of waves 4|8|12 (0: by launch size)
    // 8 lines per pair of waves, or 12 (60 instead of 40 useful lanes per load instruction: a wave 15 % longer) where that saves a
    // ROUND of waves: the kernel keeps a SIMD's issue slots 43-80 % busy, so W waves on S SIMDs last ceil(W / S) rounds whatever the
    // registers would allow (HISTORY R5.19: 136^3 -- 4624 lines = 1156 waves at 8 lines per pair -- 0.166 ms against 0.092 at 128^3;
    // with 12 lines per pair 772 waves, 0.121 ms).  Single systems of 4097 ... 6144 lines per colour and batched launches (several waves
    // per SIMD either way: 16 128 lines 4 -> 3 rounds) take 12.  The lane mapping does not touch a line's arithmetic.
    int th_lines_per_pair(const LineArgs<T>& a) const {
        if (th_lpw == 4 || th_lpw == 8 || th_lpw == 12) return th_lpw;
        const i64 lines = a.nA[0] * a.nB2[0] * (i64)nsys, simds = (i64)simd_count();
        const i64 r8 = (2 * ((lines + 7) / 8) + simds - 1) / simds, r12 = (2 * ((lines + 11) / 12) + simds - 1) / simds;
        return (23 * r12 < 20 * r8) ? 12 : 8;
    }
    mutable int cu_count = 0;
    int simd_count() const {
        if (cu_count == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
            cu_count = v;
        }
        return cu_count * 4;
    }
    int force_lpw = (int)LAB_ENV("EMG3D_LPW", 0);                       // k_line_sweep_rp: lines per wave 4|8|12 (0: by size)
    bool use_graph = !(getenv("EMG3D_GRAPH") && getenv("EMG3D_GRAPH")[0] == '0');    // replay captured cycles (0: eager launches)
    // verb = 5 of the reference (solver.py:502-578): the residual norm after every smoothing call of every level.  With
    // `trace` on the cycles run eagerly; each smoothing call is followed by a norm-only residual into trace_norms and a
    // record (iteration and cycmax of the level's loop, level, grid, kind 0 coarsest / 1 pre- / 2 post-smoothing).
    bool trace = false;
    struct TraceRec { int it, level, cycmax, kind; i64 n[3]; };
    std::vector<TraceRec> trace_recs;
    static const int TRACE_MAX = 16384;
    int trace_dropped = 0;      // smoothing calls beyond TRACE_MAX since the last emg3d_mg_get_trace (reported there)
    double* trace_norms = nullptr;
    double* norm_out = nullptr;             // where residual(mode 2) puts its norms (nullptr: norms)
    std::map<int, hipGraphExec_t> graphs;
    std::map<int, std::pair<int, int>> graph_home;      // per captured cycle: Level::e_home of level 0 on entry / on exit
    std::map<int, int> graph_seen;
    bool dry = false;           // dry run: allocate/prepare only, launch nothing
    bool use_twist = LAB_ENV("EMG3D_TWIST", 1) != 0;                    // two-sided factorisation below twist_max_lines
    i64 twist_max_lines_env = LAB_ENV("EMG3D_TWIST_MAX", 0);            // 0: q_min_lines()
    i64 twist_max_lines() const { return twist_max_lines_env > 0 ? twist_max_lines_env : q_min_lines(); }
    // ---- launch-shape thresholds in units of the DEVICE (256 CUs = 1024 SIMDs on MI355X; the literals of rounds 2-5 in brackets) ----
    // A colour launch of the chain kernels is made of waves that all last the same time: W waves on S SIMDs take ceil(W / S) rounds
    // (HISTORY R5.19), so every "how many lines" threshold is a number of waves per SIMD:
    //   q_min_lines      8 S  [8192]  lines per colour from which the quad kernel serves: one wave per SIMD at 8 lines per wave
    //   qpl_few_lines      S  [1024]  up to here the scan kernel serves lines of any length: one single-line wave per SIMD
    //   tha_min_lines  1.07 S [1100]  measured crossover of the affine kernel against the scan kernel on 33..64-block lines
    //   tha_big_lines   8 CUs [2048]  65..128-block lines in the affine kernel: one workgroup of 8 lines per CU, ONE round
    //   qdesc_max      8.8 S  [9000]  threads of a colour launch up to which the descriptor table is cheaper than the arithmetic
    i64 simds() const { return (i64)simd_count(); }
    int tw_stages = (int)LAB_ENV("EMG3D_TW_STAGES", 0);                 // register prefetch depth of the two-sided kernels (0: 3)
    bool log_launches = getenv("EMG3D_LOG") != nullptr;                 // one line per sweep launch on stderr
    int xcd_map = (int)LAB_ENV("EMG3D_XCD", 1);                         // XCD-aware workgroup -> line map
    // k_residual: block map and node planes per thread (stencil.hpp; every setting gives identical results): levels of
    // >= 1 M cells run k_residual_zm with 4 planes per thread (8 from 8 M cells x systems on) on y-strips per XCD, smaller
    // levels the plain kernel on z-slabs per XCD; launches of < 16 blocks per plane keep the plain map.
    int res_xcd = (int)LAB_ENV("EMG3D_RES_XCD", -1);                    // -1: by kernel (zm: 2, plain: 1)
    int res_xcd_min = (int)LAB_ENV("EMG3D_RES_XCD_MIN", 16);
    int res_kz = res_kz_env();                                          // 0: by size
    i64 res_zm_min = LAB_ENV("EMG3D_RES_ZM_MIN_CELLS", (i64)1 << 20);
    static int res_kz_env() {
        const int k = (int)LAB_ENV("EMG3D_RES_KZ", 0);
        return (k == 0 || k == 2 || k == 4 || k == 8 || k == 16) ? k : 1;
    }
    bool skip_idempotent = LAB_ENV("EMG3D_SKIP_IDEMPOTENT", 1) != 0;    // colour mode: skip the repeated colour at turn-arounds
    // The order in which the sweeps of a smoothing call visit the four line colours c = cP + 2 cQ: the odd ("backward")
    // sweeps 0,3,2,1, the even ("forward") ones 1,3,0,2 -- two sweeps visit 0,3,2,1,(1),3,0,2: the colour at the turn-around
    // would be solved twice in a row and is skipped (skip_idempotent).  Chosen by measurement over the 24 x 24 pairs of
    // orders (profiles/HISTORY.md A.12): forward 0,1,2,3 / backward 3,2,1,0, the mirror image of the reference's
    // lexicographic back-and-forth, is the slowest of all (0.417 mean reduction per cycle on six problems; this schedule
    // 0.356; the 128^3 F-cycle bench problem 9 -> 7 cycles at the same 7 colour passes per two sweeps).  The oracle's
    // colour twin uses the same tables.  EMG3D_COLOUR_ORDER / EMG3D_COLOUR_ORDER_B=<4 digits> (lab): other orders.
    int colour_perm[4] = {1, 3, 0, 2};
    int colour_perm_b[4] = {0, 3, 2, 1};        // as visited
    // point smoother: 8 colours c = cx + 2 cy + 4 cz, visited 0..7 in EVERY sweep (plain multi-colour Gauss-Seidel): the
    // reversed pair 7..0 / 0..7 of rounds 1-3 needed 7 / 8 / 13 cycles where this needs 5 / 6 / 9, any order repeated
    // unchanged does as well (profiles/r03_point_order.txt, HISTORY.md A.12).  EMG3D_POINT_ORDER / _B (lab): other orders.
    int point_perm[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    int point_perm_b[8] = {0, 1, 2, 3, 4, 5, 6, 7};
#ifdef EMG3D_LAB
    void read_colour_perm() {
        const char* e = getenv("EMG3D_COLOUR_ORDER");
        if (e && strlen(e) == 4) for (int k = 0; k < 4; ++k) { colour_perm[k] = (e[k] - '0') & 3; colour_perm_b[3 - k] = colour_perm[k]; }
        e = getenv("EMG3D_COLOUR_ORDER_B");
        if (e && strlen(e) == 4) for (int k = 0; k < 4; ++k) colour_perm_b[k] = (e[k] - '0') & 3;
        e = getenv("EMG3D_POINT_ORDER");
        if (e && strlen(e) == 8) for (int k = 0; k < 8; ++k) { point_perm[k] = (e[k] - '0') & 7; point_perm_b[7 - k] = point_perm[k]; }
        e = getenv("EMG3D_POINT_ORDER_B");
        if (e && strlen(e) == 8) for (int k = 0; k < 8; ++k) point_perm_b[k] = (e[k] - '0') & 7;
    }
#endif
    // sweeps on parity-split working copies (the lines of one colour contiguous in memory): 0 never, 1 every level and
    // ordering, 2 (default) colour-ordered levels of >= split_min_cells
    int use_split = (int)LAB_ENV("EMG3D_SPLIT", 2);
    i64 split_min_cells = LAB_ENV("EMG3D_SPLIT_MIN_CELLS", 2000000);
    // quad-per-line chain kernel (smooth_qc.hpp): 1 (default) on launches of >= q_min_lines lines per colour (bandwidth
    // bound: 256^3 level 0), 2 wherever a lane-group kernel would serve, 0 never
    int use_q = (int)LAB_ENV("EMG3D_Q", 1);
    i64 q_min_lines_env = LAB_ENV("EMG3D_Q_MIN_LINES", 0);              // 0: 8 lines per wave on every SIMD
    i64 q_min_lines() const { return q_min_lines_env > 0 ? q_min_lines_env : 8 * simds(); }
    // register prefetch depth of k_line_sweep_qc: 0 = by the launch -- 2 stages at 16 lines per wave and at most one wave per SIMD (210 registers; the 3-stage
    // instantiation there is 322 registers with 84 / 310 AGPR writes / reads in its loop bodies, i.e. prefetched values that are
    // waited for when they are parked), 3 stages below (level 1 of a 256^3 cycle: 8 lines per wave); lab: 2 | 3 force one.
    // 256^3, same box, alternating (profiles/r05_qstages_ab.txt): launch 731 -> 713 us dense, 650 -> 637 dipole, V-cycle 30.88 -> 30.57 ms;
    // 2 stages everywhere: the launch the same, the cycle +0.15 ms (level 1).
    int q_stages = (int)LAB_ENV("EMG3D_Q_STAGES", 0);
    // (384^3, 36.9 k lines per colour = 2.2 waves per SIMD: 3 stages again, 119.7 / 121.3 against 123.0 / 122.3 ms per V-cycle,
    // profiles/r05_qstages_ab.txt: the two-stage instantiation pays where a launch is ONE wave per SIMD)
    int q_stages_for(int lpw, i64 nmax) const {
        return q_stages == 2 || q_stages == 3 ? q_stages : ((lpw == 16 && nmax * nsys > 15 * simds() && nmax * nsys <= 16 * simds()) ? 2 : 3);
    }
    int use_zsep = (int)LAB_ENV("EMG3D_ZSEP", 1);                       // lab: 0 = always read zeta
    int q_tile = (int)LAB_ENV("EMG3D_Q_TILE", 0);                       // lab: switches of in-kernel instrumentation (LineArgs::tile; 256: timestamps of k_line_sweep_tha)
    int q_lpw = (int)LAB_ENV("EMG3D_Q_LPW", 0);                         // lines per wave 16|8|4|2 (0: by launch size)
    // quad-per-block scan kernel (smooth_qpl.hpp): direction mask; lines of qpl_min_nl .. qpl_max_nl blocks (any length
    // <= 256 when a colour has <= qpl_few_lines lines, and in lexicographic order); two blocks per quad from qpl_m2_min on
    int use_qpl = (int)LAB_ENV("EMG3D_QPL", 7);
    i64 qpl_min_nl = LAB_ENV("EMG3D_QPL_MIN", 2);
    i64 qpl_max_nl = LAB_ENV("EMG3D_QPL_MAX_NL", 64);
    // (32 since round 4: with the launch prologues trimmed, the 32-block levels of a 128^3 F-cycle -- ~1000 lines per colour -- do
    // better with one wave per SIMD and two blocks per quad than with two waves per SIMD: cycle 8.75 / 8.72 -> 8.63 / 8.65 ms;
    // from 16 blocks on: 8.78 / 8.76; profiles/r04_qpl_m2_ab.txt)
    i64 qpl_m2_min = LAB_ENV("EMG3D_QPL_M2", 32);
    // chain form of the scan kernel (smooth_qpl.hpp CH) on lines of at most this many quads (0: never).  4-block lines: the 168 such
    // launches of a 128^3 F-cycle 5.27 -> 4.86 us (three DPP-fed steps against two Kogge-Stone steps through LDS); 8-block lines
    // (seven steps, lane shuffles across the row boundary) 5.78 -> 6.16 us: 4 (HISTORY R6.3, profiles/r06_qpl_chain_ab.txt)
    int qpl_chain_seg = (int)LAB_ENV("EMG3D_QPL_CHAIN", 4);
    i64 qpl_few_lines_env = LAB_ENV("EMG3D_QPL_FEW", 0);                // 0: one single-line wave per SIMD
    i64 qpl_few_lines() const { return qpl_few_lines_env > 0 ? qpl_few_lines_env : simds(); }
    i64 qpl_max_lines = LAB_ENV("EMG3D_QPL_MAX", (i64)1 << 40);

#ifdef EMG3D_LAB
    MG() { read_colour_perm(); }
#else
    MG() {}
#endif

    ~MG() override {
        if (!stream && !side && allocs.empty() && !stage) return;       // (a shape-only object, emg3d_sweep_plan: no HIP call)
        hipSetDevice(device);
        if (stream) hipStreamSynchronize(stream);
        if (side) hipStreamSynchronize(side);
        drop_graphs();
        for (auto& pn : allocs)
            if (!DevicePool::get().give(device, pn.first, pn.second, tag_of(pn.first))) hipFree(pn.first);
        if (stage && !DevicePool::get().give(-1 - device, stage, STAGE_BYTES)) hipHostFree(stage);
        if (ev_norm) hipEventDestroy(ev_norm);
        if (ev_prep) hipEventDestroy(ev_prep);
        if (broken) {       // (streams of a handle that failed half-way -- possibly inside a capture -- are not passed on)
            if (side) (void)hipStreamDestroy(side);
            if (own_stream && stream) (void)hipStreamDestroy(stream);
            (void)hipGetLastError();
            return;
        }
        if (side) DevicePool::get().give_stream(device, side);
        if (own_stream && stream) DevicePool::get().give_stream(device, stream);
    }

    // A handle makes ~500 device allocations (levels x work copies x factor caches), most of them small:
    // hipMalloc/hipFree cost tens of microseconds each, so requests up to ARENA_CHUNK/4 are carved out of
    // 32 MiB chunks (256-byte aligned, freed together with the handle); larger ones get their own hipMalloc.
    static constexpr size_t ARENA_CHUNK = (size_t)32 << 20;
    char* arena_cur = nullptr;
    size_t arena_left = 0;
    // role tags of blocks that were placed (DevicePool): block -> tag; want_tag / got_tag: the role the next dalloc asks the pool for
    std::map<void*, int> block_tag;
    int want_tag = 0;
    bool got_tag = false;
    int tag_of(void* p) const { auto it = block_tag.find(p); return it == block_tag.end() ? 0 : it->second; }
    void* raw_alloc(size_t nb) {
        if (broken) return nullptr;             // (one failure is reported; the handle asks for nothing more)
        void* p = DevicePool::get().take(device, nb, want_tag, &got_tag);
        if (!p) {
            hipError_t st = hipMalloc(&p, nb);
            if (st != hipSuccess) {     // out of memory with blocks parked in the pool: release them and retry once
                (void)hipGetLastError();
                if (DevicePool::get().release_all() > 0) st = hipMalloc(&p, nb);
            }
            if (st != hipSuccess) {
                (void)hipGetLastError();
                if (alloc_quiet) return nullptr;        // (try_alloc: optional memory)
                err = (int)st; broken = true;
                fprintf(stderr, "[emg3d_hip] hipMalloc(%zu) failed: %s\n", nb, hipGetErrorString(st)); return nullptr;
            }
        }
        allocs.push_back({p, nb});
        return p;
    }
    template <class U>
    U* dalloc(i64 n) {
        const size_t nb = (((size_t)std::max<i64>(n, 1) * sizeof(U)) + 255) & ~(size_t)255;
        bytes += (i64)nb;
        if (nb > ARENA_CHUNK / 4) return (U*)raw_alloc(nb);
        if (nb > arena_left) {
            arena_cur = (char*)raw_alloc(ARENA_CHUNK);
            arena_left = arena_cur ? ARENA_CHUNK : 0;
            if (!arena_cur) return nullptr;
        }
        U* p = (U*)arena_cur;
        arena_cur += nb;
        arena_left -= nb;
        return p;
    }
    // Host -> device copy of a set-up array (widths, transfer weights: ~150 small arrays per hierarchy).
    // The host buffers may be temporaries, so small ones are staged through a pinned ring buffer and
    // copied asynchronously (one stream synchronisation per STAGE_BYTES instead of one per array).
    static constexpr size_t STAGE_BYTES = (size_t)2 << 20;
    static constexpr size_t STAGE_RING = STAGE_BYTES - 4096;      // the last 4 KiB: pinned landing place of the norms
    char* stage = nullptr;
    size_t stage_off = 0;
    hipEvent_t ev_norm = nullptr;
    void ensure_stage() {
        if (!stage) stage = (char*)DevicePool::get().take(-1 - device, STAGE_BYTES);
        if (!stage && hipHostMalloc((void**)&stage, STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { stage = nullptr; (void)hipGetLastError(); }
    }
    // One cycle, and while the device runs it the loop-invariant set-up of the NEXT (sc_dir, lr_dir) pair on the host
    // (hierarchy, factor kernels, graph capture: 5 ms at 128^3 -- the solver rotates through up to three pairs, so the
    // second and third cycle of a solve otherwise wait for it).  The norms come back through pinned memory behind an
    // event recorded right after the cycle; what prepare() enqueues runs after it.
    // prepare() with everything it enqueues -- uploads, coarse models, factor kernels, the graph upload -- on a second
    // stream, so that the device runs it BESIDE the cycle instead of behind it (the coarse levels of a cycle leave most
    // CUs idle); the handle's stream then waits for it before whatever comes next.  The new pair's buffers are not
    // touched by the running cycle, and what it shares with it (level-0 model, source, widths) is only read.
    hipStream_t side = nullptr;
    hipEvent_t ev_prep = nullptr;
    void prepare_aside(int g, int lr_dir) {
        // (a pair whose set-up still has to place level 0's working copies times sweeps: on the handle's own stream, behind the cycle)
        if (!prepare_on_side || placement_pending(lr_dir)) { prepare(g, lr_dir); return; }
        if (!side) side = DevicePool::get().take_stream(device);
        if (!side || (!ev_prep && hipEventCreateWithFlags(&ev_prep, hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError();
            prepare(g, lr_dir);
            return;
        }
        std::swap(stream, side);
        prepare(g, lr_dir);
        hipError_t st = hipEventRecord(ev_prep, stream);
        std::swap(stream, side);
        if (st == hipSuccess) st = hipStreamWaitEvent(stream, ev_prep, 0);
        if (st != hipSuccess) {                       // cannot order the streams: wait on the host instead
            (void)hipGetLastError();
            hipStreamSynchronize(side);
        }
    }
    int prepare_on_side = (int)LAB_ENV("EMG3D_PREPARE_SIDE", 1);
    int cycle_then_prepare(int g, int lr_dir, int ng, int nlr, double* out) {
        cycle0(g, lr_dir, 0);
        ensure_stage();
        if (!stage || nsys * sizeof(double) > 4096) {           // no pinned memory: plain order
            if (ng >= 0) prepare(ng, nlr);
            hipError_t st = d2h(out, norms, (size_t)nsys * sizeof(double));
            if (st != hipSuccess && err == 0) err = (int)st;
            return 0;
        }
        double* land = (double*)(stage + STAGE_RING);
        hipError_t st = hipMemcpyAsync(land, norms, (size_t)nsys * sizeof(double), hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess && !ev_norm) st = hipEventCreateWithFlags(&ev_norm, hipEventDisableTiming);
        if (st == hipSuccess) st = hipEventRecord(ev_norm, stream);
        if (st == hipSuccess && ng >= 0) prepare_aside(ng, nlr);
        if (st == hipSuccess) st = hipEventSynchronize(ev_norm);
        if (st != hipSuccess) { if (err == 0) err = (int)st; return 0; }
        memcpy(out, land, (size_t)nsys * sizeof(double));
        return 0;
    }
    template <class U>
    U* upload(const U* host, i64 n) {
        U* d = dalloc<U>(n);
        if (!d || n <= 0) return d;
        const size_t nb = (size_t)n * sizeof(U);
        hipError_t st;
        ensure_stage();
        if (stage && nb <= STAGE_RING / 4) {
            if (stage_off + nb > STAGE_RING) {      // (both streams: copies out of the ring may be pending on either)
                hipStreamSynchronize(stream);
                if (side) hipStreamSynchronize(side);
                stage_off = 0;
            }
            memcpy(stage + stage_off, host, nb);
            st = hipMemcpyAsync(d, stage + stage_off, nb, hipMemcpyHostToDevice, stream);
            stage_off += (nb + 63) & ~(size_t)63;
        } else {
            st = h2d(d, host, nb);
        }
        if (st != hipSuccess) err = (int)st;
        return d;
    }
    // Whole-field host <-> device copies, synchronous (the runtime pins the caller's array and DMAs from
    // it: 102 MB in 2-8 ms; staging through own pinned chunks was measured slower and is not needed).
    hipError_t h2d(void* dst, const void* src, size_t nb) {
        hipError_t st = hipMemcpyAsync(dst, src, nb, hipMemcpyHostToDevice, stream);
        return st == hipSuccess ? hipStreamSynchronize(stream) : st;
    }
    hipError_t d2h(void* dst, const void* src, size_t nb) {
        hipError_t st = hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToHost, stream);
        return st == hipSuccess ? hipStreamSynchronize(stream) : st;
    }

    void check_launch() {
        hipError_t st = hipGetLastError();
        if (st != hipSuccess && err == 0) { err = (int)st; fprintf(stderr, "[emg3d_hip] launch failed: %s\n", hipGetErrorString(st)); }
    }

    // --------------------------------------------------------------- levels
    // sizes and layouts that follow from nC alone (no device memory)
    static void shape_level(Level<T>& L) {
        L.nE = n_edges(L.nC);
        L.nCells = L.nC[0] * L.nC[1] * L.nC[2];
        L.fl = ref_field_layout(L.nC);
        L.cl = ref_cell_layout(L.nC);
        L.flT = L.fl;
        for (int c = 0; c < 3; ++c) {
            const i64 d0 = (c == 0) ? L.nC[0] : L.nC[0] + 1, d1 = (c == 1) ? L.nC[1] : L.nC[1] + 1;
            L.flT.st[c][0] = d1; L.flT.st[c][1] = 1; L.flT.st[c][2] = d0 * d1;
        }
        L.clT.st[0] = L.nC[1]; L.clT.st[1] = 1; L.clT.st[2] = L.nC[0] * L.nC[1];
    }
    std::shared_ptr<Level<T>> make_level(const std::vector<double> hh[3]) {
        auto L = std::make_shared<Level<T>>();
        for (int a = 0; a < 3; ++a) {
            L->nC[a] = (i64)hh[a].size();
            L->h_host[a] = hh[a];
            // nodes = r_[0, cumsum(h)] + origin ; centers (meshes.py:84-97)
            L->nodes[a].resize(hh[a].size() + 1);
            double cs = 0.0;
            L->nodes[a][0] = 0.0 + origin[a];
            for (size_t i = 0; i < hh[a].size(); ++i) { cs += hh[a][i]; L->nodes[a][i + 1] = cs + origin[a]; }
            L->centers[a].resize(hh[a].size());
            for (size_t i = 0; i < hh[a].size(); ++i) L->centers[a][i] = (L->nodes[a][i + 1] + L->nodes[a][i]) / 2;
            L->h[a] = upload<double>(hh[a].data(), (i64)hh[a].size());
            std::vector<double> inv(hh[a].size());
            for (size_t i = 0; i < hh[a].size(); ++i) inv[i] = 1.0 / hh[a][i];
            L->ih[a] = upload<double>(inv.data(), (i64)inv.size());
        }
        shape_level(*L);
        L->s = dalloc<T>(nsys * L->nE);
        L->e = dalloc<T>(nsys * L->nE);
        L->r = dalloc<T>(nsys * L->nE);
        return L;
    }

    // Build the transfer operators + child model of `L` for current sc_dir `sc`.
    std::shared_ptr<Level<T>> make_child(Level<T>& L, Transfer& X, int sc) {
        X.sc = sc;
        sc_axes(sc, X.co);
        std::vector<double> ch[3];
        for (int a = 0; a < 3; ++a) {
            if (X.co[a]) {   // ch = diff(nodes[::2]), solver.py:859-861
                const i64 n = L.nC[a] / 2;
                ch[a].resize(n);
                for (i64 i = 0; i < n; ++i) ch[a][i] = L.nodes[a][2 * i + 2] - L.nodes[a][2 * i];
            } else {
                // np.diff(nodes[::1])
                ch[a].resize(L.nC[a]);
                for (i64 i = 0; i < L.nC[a]; ++i) ch[a][i] = L.nodes[a][i + 1] - L.nodes[a][i];
            }
        }
        auto C = make_level(ch);
        // model (solver.py:874-884), aliasing preserved
        const int blocks = (int)((C->nCells + EMG_BLOCK - 1) / EMG_BLOCK);
        C->eta[0] = dalloc<T>(C->nCells);
        MG_LAUNCH(k_restrict_model<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, C->eta[0],
                           (const T*)L.eta[0], C->nC[0], C->nC[1], C->nC[2], L.nC[0], L.nC[1], X.co[0], X.co[1], X.co[2]);
        for (int c = 1; c < 3; ++c) {
            if (eta_alias[c]) { C->eta[c] = C->eta[0]; continue; }
            C->eta[c] = dalloc<T>(C->nCells);
            MG_LAUNCH(k_restrict_model<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, C->eta[c],
                               (const T*)L.eta[c], C->nC[0], C->nC[1], C->nC[2], L.nC[0], L.nC[1], X.co[0], X.co[1], X.co[2]);
        }
        C->zeta = dalloc<double>(C->nCells);
        MG_LAUNCH(k_restrict_model<double>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, C->zeta,
                           (const double*)L.zeta, C->nC[0], C->nC[1], C->nC[2], L.nC[0], L.nC[1], X.co[0], X.co[1], X.co[2]);
        check_launch();
        // restriction weights (solver.py:1787-1838) and prolongation weights
        for (int a = 0; a < 3; ++a) {
            if (X.co[a]) {
                const i64 n = C->nC[a] + 1;
                std::vector<double> wl(n), w0(n), wr(n);
                restrict_weights_host(L.nodes[a].data(), L.centers[a].data(), L.h_host[a].data(), L.nC[a],
                                      C->nodes[a].data(), C->centers[a].data(), C->h_host[a].data(), n,
                                      wl.data(), w0.data(), wr.data());
                X.w[a][0] = upload<double>(wl.data(), n);
                X.w[a][1] = upload<double>(w0.data(), n);
                X.w[a][2] = upload<double>(wr.data(), n);
            }
            std::vector<int> idx; std::vector<double> wt;
            prolong_weights_host(C->nodes[a], L.nodes[a], idx, wt);
            X.pidx[a] = upload<int>(idx.data(), (i64)idx.size());
            X.pwt[a] = upload<double>(wt.data(), (i64)wt.size());
        }
        return C;
    }

    // ---- another frequency on the same handle (handles created from sigma*V, emg3d_mg_create_sv) ----
    // eta = s mu_0 sigma V changes with the frequency, nothing else does: grids, transfer weights, work buffers and the
    // captured launch graphs (they hold pointers, not values) stay.  Recomputed with the kernels and in the order of a
    // fresh handle, so the results are those of a fresh handle bit for bit: level-0 eta, the coarse models of every
    // hierarchy built so far, transposed model copies, every cached line factorisation.
    double* sv[3] = {nullptr, nullptr, nullptr};
    double* volw = nullptr;        // != nullptr: sv[] holds the conductivities, eta = (b V) sigma (k_eta_vs: VolumeModel's rounding)
    double* epsr = nullptr;        // != nullptr (with volw): relative permittivities, eta = (b V) (sigma - c eps_r) (k_eta_vs_eps)
    double seps0 = 0.0;            // c = s eps_0 (Laplace domain) resp. Im(s) eps_0 (frequency domain) of the current frequency
    static double imag_or_real(double x) { return x; }
    static double imag_or_real(c128 x) { return x.im; }
    void form_eta(Level<T>& L0, T smu0) {
        const unsigned blocks = (unsigned)std::min<i64>((L0.nCells + EMG_BLOCK - 1) / EMG_BLOCK, 4096);
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && eta_alias[c]) continue;
            if (volw && epsr)
                MG_LAUNCH(k_eta_vs_eps<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, L0.eta[c], (const double*)volw,
                                   (const double*)sv[c], (const double*)epsr, imag_or_real(smu0), seps0, L0.nCells);
            else if (volw)
                MG_LAUNCH(k_eta_vs<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, L0.eta[c], (const double*)volw,
                                   (const double*)sv[c], imag_or_real(smu0), L0.nCells);
            else
                MG_LAUNCH(k_scale_real_to<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, L0.eta[c], (const double*)sv[c], smu0, L0.nCells);
        }
    }
    void restrict_eta(Level<T>& L, const Transfer& X, Level<T>& C) {
        const int blocks = (int)((C.nCells + EMG_BLOCK - 1) / EMG_BLOCK);
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && eta_alias[c]) continue;
            MG_LAUNCH(k_restrict_model<T>, dim3(blocks), dim3(EMG_BLOCK), 0, stream, C.eta[c],
                               (const T*)L.eta[c], C.nC[0], C.nC[1], C.nC[2], L.nC[0], L.nC[1], X.co[0], X.co[1], X.co[2]);
        }
    }
    int set_smu0(T smu0) {
        if (!sv[0]) return -7;
        Level<T>& L0 = *lv0;
        form_eta(L0, smu0);
        for (auto& kv : hier) {
            Hierarchy<T>& H = kv.second;
            for (size_t l = 0; l + 1 < H.lv.size(); ++l) restrict_eta(*H.lv[l], H.tr[l], *H.lv[l + 1]);
        }
        std::vector<Level<T>*> seen;
        auto refresh = [&](Level<T>& L) {
            for (auto* p : seen) if (p == &L) return;
            seen.push_back(&L);
            if (L.zetaT) {
                transpose_xy(L.etaT[0], (const T*)L.eta[0], L.nC[0], L.nC[1], L.nC[2], true, 0);
                for (int c = 1; c < 3; ++c)
                    if (L.eta[c] != L.eta[0]) transpose_xy(L.etaT[c], (const T*)L.eta[c], L.nC[0], L.nC[1], L.nC[2], true, 0);
            }
            for (int d = 0; d < 3; ++d) if (L.fac[d]) compute_factor(L, d);
        };
        refresh(L0);
        for (auto& kv : hier) for (auto& l : kv.second.lv) if (l) refresh(*l);
        check_launch();
        return 0;
    }

    // Level 0 of a model without magnetic permeabilities: zeta is the cell volume (hx hy) hz.  Checked bit for bit on the
    // device; the level-0 sweep kernels then form zeta from the width vectors instead of reading it (smooth_qc.hpp).
    void check_zeta() {
        Level<T>& L = *lv0;
        int* flag = dalloc<int>(1);
        if (!flag) return;
        hipMemsetAsync(flag, 0, sizeof(int), stream);
        MG_LAUNCH(k_zeta_is_volume, dim3((unsigned)std::min<i64>((L.nCells + EMG_BLOCK - 1) / EMG_BLOCK, 4096)), dim3(EMG_BLOCK),
                           0, stream, (const double*)L.zeta, (const double*)L.h[0], (const double*)L.h[1], (const double*)L.h[2],
                           L.nC[0], L.nC[1], L.nC[2], flag);
        int host = 1;
        if (hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); host = 1; }
        L.zeta_sep = (host == 0);
    }

    // Hierarchy for global sc_dir g: levels 0..clevel[g] (solver.py:480, 524, 551).
    Hierarchy<T>& hierarchy(int g) {
        auto it = hier.find(g);
        if (it != hier.end()) return it->second;
        Hierarchy<T> H;
        H.lv.push_back(lv0);              // level 0 (arrays + caches) is shared by all hierarchies
        for (int lev = 0; lev < clevel[g]; ++lev) {
            Level<T>& L = *H.lv.back();
            const int sc = current_sc_dir(g, L.nC);
            H.tr.emplace_back();
            H.lv.push_back(make_child(L, H.tr.back(), sc));
        }
        hier[g] = H;
        return hier[g];
    }

    // ------------------------------------------- Krylov vector workspace
    // nE-sized device vectors addressed by index; -1 = level-0 source s, -2 = level-0 field e.
    std::vector<T*> vecs;
    double* dot_partials = nullptr;
    double* dot_out = nullptr;
    static const int DOT_BLOCKS = 1024;
    int vec_alloc(int n) {
        while ((int)vecs.size() < n) {
            T* v = try_alloc<T>(lv0->nE);       // (optional memory: the caller falls back to the host iteration; the handle stays usable)
            if (!v) return (int)hipErrorOutOfMemory;                   // (nothing pushed: vec() keeps answering nullptr)
            hipMemsetAsync(v, 0, (size_t)lv0->nE * sizeof(T), stream);
            vecs.push_back(v);
        }
        if (!dot_partials) { dot_partials = dalloc<double>(2 * DOT_BLOCKS); dot_out = dalloc<double>(2); }
        return broken ? (int)hipErrorOutOfMemory : err;
    }
    T* vec(int id) {
        if (id == -1) return sel_s();
        if (id == -2) return sel_e();
        if (id < 0 || id >= (int)vecs.size()) return nullptr;
        return vecs[id];
    }
    unsigned vec_grid() const { return (unsigned)std::min<i64>((lv0->nE + EMG_BLOCK - 1) / EMG_BLOCK, 4096); }
    void touched(int id) {      // the level-0 source changed: its working copies are stale
        if (id == -1) { source_changed(); lv0->s_dense = true; }      // a dense vector: no point in scanning it for zeros
    }
    int vec_copy(int dst, int src) {
        T *d = vec(dst), *s_ = vec(src);
        if (!d || !s_) return -2;
        if (d != s_) hipMemcpyAsync(d, s_, (size_t)lv0->nE * sizeof(T), hipMemcpyDeviceToDevice, stream);
        touched(dst);
        return 0;
    }
    int vec_axpy(int y, T alpha, int x) {
        T *py = vec(y), *px = vec(x);
        if (!py || !px || py == px) return -2;
        MG_LAUNCH(k_axpy<T>, dim3(vec_grid()), dim3(EMG_BLOCK), 0, stream, py, (const T*)px, alpha, lv0->nE);
        touched(y);
        return 0;
    }
    int vec_scale(int y, T alpha) {
        T* py = vec(y);
        if (!py) return -2;
        MG_LAUNCH(k_scale<T>, dim3(vec_grid()), dim3(EMG_BLOCK), 0, stream, py, alpha, lv0->nE);
        touched(y);
        return 0;
    }
    int vec_dot(int a, int b, double out[2]) {     // <a, b> with a conjugated
        T *pa = vec(a), *pb = vec(b);
        if (!pa || !pb || !dot_partials) return -2;
        MG_LAUNCH(k_dot_partials<T>, dim3(DOT_BLOCKS), dim3(EMG_BLOCK), 0, stream, (const T*)pa, (const T*)pb,
                           lv0->nE, dot_partials);
        MG_LAUNCH(k_sum_pairs, dim3(1), dim3(EMG_BLOCK), 0, stream, (const double*)dot_partials, (i64)DOT_BLOCKS, dot_out);
        hipError_t st = hipMemcpyAsync(out, dot_out, 2 * sizeof(double), hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess) st = hipStreamSynchronize(stream);
        return st == hipSuccess ? 0 : (int)st;
    }
    // dst = A src  (A = the system matrix; reference amatvec, solver.py:646-660)
    int vec_amatvec(int dst, int src) {
        T *pd = vec(dst), *ps = vec(src);
        if (!pd || !ps || pd == ps) return -2;
        Level<T>& L = *lv0;
        hipMemsetAsync(pd, 0, (size_t)L.nE * sizeof(T), stream);
        ResidualArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = L.nC[q]; a.eta[q] = L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q]; }
        a.fl = L.fl; a.r = pd; a.s = pd; a.e = ps; a.zeta = L.zeta; a.partials = nullptr; a.bt = Batch();
        const i64 plane = (L.nC[0] + 1) * (L.nC[1] + 1);
        dim3 grid((unsigned)((plane + EMG_BLOCK - 1) / EMG_BLOCK), (unsigned)(L.nC[2] + 1));
        if (!broken) residual_launch<T>(0, 1, grid, stream, a);   // pd = 0 - A ps
        MG_LAUNCH(k_negate<T>, dim3(vec_grid()), dim3(EMG_BLOCK), 0, stream, pd, L.nE);
        touched(dst);
        check_launch();
        return err;
    }

    // ------------------------------------------------------------ smoothers
    // ---- working copies: x<->y transpose and parity split ------------------
    // bt: the batch of a FIELD array (model arrays are shared by the systems: Batch())
    template <class U>
    void transpose_xy(U* dst, const U* src, i64 d0, i64 d1, i64 d2, bool to_T, int split, Batch bt = Batch()) {
        // src (d0 fastest, d1, d2) -> dst (d1 fastest, d0, d2) if to_T, else the inverse;
        // split: 1 = the d1-fastest side is parity-split, 2 = both sides are
        const i64 a0 = to_T ? d0 : d1, a1 = to_T ? d1 : d0;
        dim3 grid((unsigned)((a0 + 31) / 32), (unsigned)((a1 + 31) / 32), (unsigned)(d2 * (bt.st ? nsys : 1)));
        if (!split) MG_LAUNCH((k_transpose01<U, 0>), grid, dim3(32, 8), 0, stream, dst, src, a0, a1, (int)d2, bt);
        else if (split == 2) MG_LAUNCH((k_transpose01<U, 2>), grid, dim3(32, 8), 0, stream, dst, src, a0, a1, (int)d2, bt);
        else if (to_T) MG_LAUNCH((k_transpose01<U, 1>), grid, dim3(32, 8), 0, stream, dst, src, a0, a1, (int)d2, bt);
        else MG_LAUNCH((k_transpose01<U, -1>), grid, dim3(32, 8), 0, stream, dst, src, a0, a1, (int)d2, bt);
    }
    template <class U>
    void split_x(U* dst, const U* src, i64 d0, i64 rows, bool to_split, Batch bt = Batch()) {
        const i64 n = d0 * rows;
        const dim3 blocks((unsigned)std::min<i64>((n + EMG_BLOCK - 1) / EMG_BLOCK, 16384), bt.st ? nsys : 1);
        if (to_split) MG_LAUNCH((k_split0<U, 1>), blocks, dim3(EMG_BLOCK), 0, stream, dst, src, d0, rows, bt);
        else MG_LAUNCH((k_split0<U, -1>), blocks, dim3(EMG_BLOCK), 0, stream, dst, src, d0, rows, bt);
    }
    // ---- batched systems ------------------------------------------------------------------------------
    int nsys = 1;               // systems (right-hand sides) that run through every launch of the cycle
    int cur = 0;                // the system the single-field entry points (set/get field, source, receivers) address
    int* bmask = nullptr;       // device, nsys flags: 0 = frozen (nullptr: all active)
    // level-0 arrays of the selected system
    T* sel_s() { return lv0->s + (i64)cur * lv0->nE; }
    T* sel_e() { e_to_ref(*lv0); return lv0->e + (i64)cur * lv0->nE; }
    T* sel_r() { return lv0->r + (i64)cur * lv0->nE; }
    void source_changed() {
        lv0->sT_valid = false; lv0->sW_valid[0] = lv0->sW_valid[1] = false;
        lv0->sflag_valid[0] = lv0->sflag_valid[1] = lv0->sflag_valid[2] = false;
        lv0->s_dense = false;
    }
    // Source-free lines of level 0 (smooth_qc.hpp): flags per line direction, kept current like the source's working copies --
    // recomputed outside the captured graphs whenever the source has changed.  Batched systems: [system][line].
    int use_sflag = (int)LAB_ENV("EMG3D_SFLAG", 1);                     // lab: 0 = the sweeps always read the source
    bool sflag_on(const Level<T>& L, int dir) const {
        return use_sflag && &L == lv0.get() && order == 1 && L.fac[dir] && (L.fac_kind[dir] == 3 || L.fac_kind[dir] == 4);
    }
    void ensure_sflags(Level<T>& L, int dir) {
        if (!sflag_on(L, dir)) return;
        LineArgs<T> a;
        line_args(L, dir, a, false);
        if (!L.sflag[dir]) { L.sflag[dir] = dalloc<unsigned char>(a.nLinesTot * nsys); L.sflag_valid[dir] = false; }
        if (dry || L.sflag_valid[dir] || !L.sflag[dir]) return;
        const i64 nmax = a.nA[0] * a.nB2[0];
        if (L.s_dense) hipMemsetAsync(L.sflag[dir], 1, (size_t)(a.nLinesTot * nsys), stream);
        else if (nmax > 0)
            MG_LAUNCH(k_source_line_flags<T>, dim3((unsigned)((nmax + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK), 4, (unsigned)nsys),
                               dim3(EMG_LINE_BLOCK), 0, stream, a, (const T*)L.s, L.fl, L.sflag[dir], L.nE);
        L.sflag_valid[dir] = true;
    }
    // Give a large allocation back (only whole hipMalloc blocks; arena pieces stay until the handle goes).
    void release(void* p) {
        for (size_t i = 0; i < allocs.size(); ++i)
            if (allocs[i].first == p) {
                bytes -= (i64)allocs[i].second;
                block_tag.erase(p);             // (given back untagged: a candidate that lost, an array of another shape)
                if (!DevicePool::get().give(device, p, allocs[i].second)) hipFree(p);
                allocs.erase(allocs.begin() + (long)i);
                return;
            }
    }
    // n systems per launch.  Only before the first cycle (no hierarchy, no captured graph yet): the level-0
    // arrays are re-allocated as [n][nE]; the model and everything derived from it is shared.
    int set_batch(int n) {
        if (n < 1 || n > 64) return -2;
        if (!hier.empty() || !graphs.empty() || lv0->eT || lv0->eW[0] || lv0->eW[1] || !vecs.empty()) return -6;
        if (n == nsys) return 0;
        Level<T>& L = *lv0;
        hipStreamSynchronize(stream);
        release(L.s); release(L.e); release(L.r);
        nsys = n; cur = 0;
        L.s = dalloc<T>(nsys * L.nE); L.e = dalloc<T>(nsys * L.nE); L.r = dalloc<T>(nsys * L.nE);
        if (!L.s || !L.e || !L.r) return err ? err : -3;
        hipMemsetAsync(L.s, 0, (size_t)(nsys * L.nE) * sizeof(T), stream);
        hipMemsetAsync(L.e, 0, (size_t)(nsys * L.nE) * sizeof(T), stream);
        hipMemsetAsync(L.r, 0, (size_t)(nsys * L.nE) * sizeof(T), stream);
        bmask = nullptr; n_partials = 0;
        if (nsys > 1) {
            bmask = dalloc<int>(nsys);
            std::vector<int> ones((size_t)nsys, 1);
            hipMemcpyAsync(bmask, ones.data(), (size_t)nsys * sizeof(int), hipMemcpyHostToDevice, stream);
        }
        source_changed();
        hipError_t st = hipStreamSynchronize(stream);
        return st == hipSuccess ? 0 : (int)st;
    }
    int set_mask(const int* host) {
        if (nsys == 1) return 0;
        hipError_t st = hipMemcpyAsync(bmask, host, (size_t)nsys * sizeof(int), hipMemcpyHostToDevice, stream);
        if (st == hipSuccess) st = hipStreamSynchronize(stream);
        source_changed();       // a system that was frozen while a working copy of the source was refreshed
        return st == hipSuccess ? 0 : (int)st;
    }
    Batch batch(const Level<T>& L) const { return Batch{nsys > 1 ? L.nE : 0, nsys > 1 ? bmask : nullptr, nsys}; }
    dim3 bgrid(unsigned g) const { return dim3(g * (unsigned)nsys, 1, 1); }      // line sweeps: EMG_SWEEP_WG
    dim3 bgrid_y(unsigned g) const { return dim3(g, (unsigned)nsys, 1); }
    // field: reference layout <-> working copy w (0: transposed + y-split, 1: x-split,
    // -1: plain transpose); w = 2: working copy 1 -> working copy 0 (to_work) or back.  all: frozen systems too.
    void convert_field(Level<T>& L, T* dst, const T* src, int w, bool to_work, bool all = false) {
        Batch bt = batch(L);
        if (all) bt.mask = nullptr;
        if (w == -1) {      // plain x<->y transposition: the three components in one launch
            FieldTransArgs t;
            i64 m0 = 0, m1 = 0;
            for (int c = 0; c < 3; ++c) {
                const i64 d0 = (c == 0) ? L.nC[0] : L.nC[0] + 1, d1 = (c == 1) ? L.nC[1] : L.nC[1] + 1,
                          d2 = (c == 2) ? L.nC[2] : L.nC[2] + 1;
                t.off[c] = L.fl.off[c]; t.a0[c] = to_work ? d0 : d1; t.a1[c] = to_work ? d1 : d0; t.nz[c] = (int)d2;
                m0 = std::max(m0, t.a0[c]); m1 = std::max(m1, t.a1[c]);
            }
            dim3 grid((unsigned)((m0 + 31) / 32), (unsigned)((m1 + 31) / 32), (unsigned)((t.nz[0] + t.nz[1] + t.nz[2]) * nsys));
            MG_LAUNCH((k_transpose01_field<T>), grid, dim3(32, 8), 0, stream, dst, src, t, bt);
            return;
        }
        for (int c = 0; c < 3; ++c) {
            const i64 d0 = (c == 0) ? L.nC[0] : L.nC[0] + 1, d1 = (c == 1) ? L.nC[1] : L.nC[1] + 1,
                      d2 = (c == 2) ? L.nC[2] : L.nC[2] + 1;
            if (w == 1) split_x(dst + L.fl.off[c], src + L.fl.off[c], d0, d1 * d2, to_work, bt);
            else transpose_xy(dst + L.fl.off[c], src + L.fl.off[c], d0, d1, d2, to_work, w == 2 ? 2 : 1, bt);
        }
    }
    void ensure_transposed_model(Level<T>& L) {
        if (L.zetaT) return;
        L.eT = dalloc<T>(nsys * L.nE); L.sT = dalloc<T>(nsys * L.nE);
        L.etaT[0] = dalloc<T>(L.nCells);
        transpose_xy(L.etaT[0], (const T*)L.eta[0], L.nC[0], L.nC[1], L.nC[2], true, 0);
        for (int c = 1; c < 3; ++c) {
            if (L.eta[c] == L.eta[0]) { L.etaT[c] = L.etaT[0]; continue; }
            L.etaT[c] = dalloc<T>(L.nCells);
            transpose_xy(L.etaT[c], (const T*)L.eta[c], L.nC[0], L.nC[1], L.nC[2], true, 0);
        }
        L.zetaT = dalloc<double>(L.nCells);
        transpose_xy(L.zetaT, (const double*)L.zeta, L.nC[0], L.nC[1], L.nC[2], true, 0);
        L.sT_valid = false;
    }
    void ensure_work(Level<T>& L, int w) {
        if (L.eW[w]) return;
        if (&L == lv0.get() && place_applies(L) && !placed[w]) {
            // a block that an earlier handle of this process searched for this role (DevicePool tags): no second search
            want_tag = 1 + w;
            L.eW[w] = dalloc<T>(nsys * L.nE);
            want_tag = 0;
            if (L.eW[w] && got_tag) { placed[w] = true; block_tag[L.eW[w]] = 1 + w; place_rec[w] = PlaceRec(); place_rec[w].reused = 1; }
        } else
        L.eW[w] = dalloc<T>(nsys * L.nE);
        L.sW[w] = dalloc<T>(nsys * L.nE);
        L.zetaW[w] = dalloc<double>(L.nCells);
        if (w == 1) split_x(L.zetaW[w], (const double*)L.zeta, L.nC[0], L.nC[1] * L.nC[2], true);
        else transpose_xy(L.zetaW[w], (const double*)L.zeta, L.nC[0], L.nC[1], L.nC[2], true, 1);
        L.sW_valid[w] = false;
    }
    // does the row-parallel kernel (32-bit offsets) apply to this level?
    // The lane-group kernels address with a uniform base and 32-bit per-lane byte offsets: (i) the field arrays, (ii) a plane of
    // the factor, (iii) zeta must each stay below 4 GiB.  wide_fits: (ii) and (iii) only.
    bool wide_fits(const Level<T>& L) const {
        const i64 lim = (i64)1 << 32;
        i64 mx = 0;
        for (int d = 0; d < 3; ++d) {
            const int P = (d == 0) ? 1 : 0, Q = (d == 2) ? 1 : 2;
            mx = std::max(mx, (L.nC[P] - 1) * (L.nC[Q] - 1));
        }
        return sweep_kernel == 0 && mx * 15 * (i64)sizeof(T) < lim && L.nCells * 8 < lim;
    }
    bool rp_fits(const Level<T>& L) const {
        return wide_fits(L) && L.nE * (i64)sizeof(T) < ((i64)1 << 32) && !q_big_force(L);
    }
    // Fields of 4 GiB and more (complex: from ~445^3 cells on; 512^3 = 6.4 GB per field): the quad-per-line kernel with 64-bit
    // field offsets (k_line_sweep_qc<..., BIG>) serves, on the split working copies like every other large level, where all
    // three line directions have the lines for it (>= q_min_lines() per colour, i.e. 8 per wave on every SIMD: the 16-line instantiation
    // at the balanced number of lines per wave); everything else of the cycle
    // (residual, transfers, conversions) is 64-bit throughout.  Other shapes keep the thread-per-line kernel.
    // EMG3D_Q_BIG=1 (lab): the 64-bit variant on levels that would fit 32 bits (parity tests at small sizes).
    int q_big_lab = (int)LAB_ENV("EMG3D_Q_BIG", 0);
    bool q_big_lines(const Level<T>& L) const {
        for (int d = 0; d < 3; ++d) {
            const int P = (d == 0) ? 1 : 0, Q = (d == 2) ? 1 : 2;
            if ((L.nC[P] / 2) * (L.nC[Q] / 2) < std::max<i64>(q_min_lines(), 1)) return false;
        }
        return order == 1 && use_q >= 1;
    }
    bool q_big_force(const Level<T>& L) const { return q_big_lab && q_big_lines(L) && wide_fits(L); }
    bool q_big(const Level<T>& L) const {
        return wide_fits(L) && q_big_lines(L) && (q_big_lab || L.nE * (i64)sizeof(T) >= ((i64)1 << 32));
    }
    bool split_on(const Level<T>& L) const {
        return (use_split == 1 || (use_split == 2 && order == 1 && L.nCells >= split_min_cells)) && (rp_fits(L) || q_big(L));
    }
    // Level 0 with split working copies: the field STAYS in the x-split copy eW[1] between the sweeps -- the residual and
    // the prolongation address it there (ResidualArgs::xs, ProlongArgs::fxs), the x-line sweeps convert eW[1] <-> eW[0]
    // -- and returns to the reference layout only when something outside the cycle asks for it (sel_e).  Saves the
    // un-split / split passes around every smoothing step (5 % of a 256^3 V-cycle).  EMG3D_HOME=0 (lab): off.
    // The coarse levels with split copies never see the reference layout at all: restriction and prolongation address
    // eW[1] / sW[1] directly (RestrictArgs::cxs, ProlongArgs::cxs).
    int use_home = (int)LAB_ENV("EMG3D_HOME", 1);
    bool home_on(const Level<T>& L) const { return use_home && &L == lv0.get() && split_on(L); }
    bool home_lvl(const Level<T>& L) const { return use_home && &L != lv0.get() && split_on(L); }
    void e_to_ref(Level<T>& L) {
        if (L.e_home != 1) return;
        convert_field(L, L.e, L.eW[1], 1, false, true);
        L.e_home = 0;
    }
    void e_to_w1(Level<T>& L) {
        if (L.e_home == 1) return;
        ensure_work(L, 1);
        convert_field(L, L.eW[1], L.e, 1, true, true);
        L.e_home = 1;
    }

    // dir 0 (x-lines) runs on the transposed copies when `use_xt`.
    // Small levels: the 6-9 transposition launches cost more than strided access.
    bool xt(const Level<T>& L, int dir) const {
        return dir == 0 && use_xt && L.nCells >= xt_min_cells && !qpl(L, dir);
    }
    // Which sweep kernel serves (level, direction) -- decided when the factor is built, because the
    // kernels differ in the factor layout:
    //   k_line_sweep_qpl (scan along the line, smooth_qpl.hpp)  wherever the dependent chain of the lane-group
    //       kernels would leave SIMDs idle: lines of <= qpl_max_nl (64) blocks; lines of any length <= 256
    //       blocks when a colour has <= qpl_few_lines (1024) lines; every launch of the lexicographic order
    //       (a hyperplane holds at most min(nP, nQ)/2 lines: 128-block lines 20 instead of 96 us per launch);
    //   k_line_sweep_thm (two-sided chain on the mirrored factorisation, halves of 8 lines in a pair of waves, smooth_thm.hpp)
    //                                                             colours of < 8192 longer lines (128^3 level 0);
    //   k_line_sweep_qc  (quad per line, compact factor, smooth_qc.hpp)   colours of >= 8192 lines (256^3 levels 0, 1);
    //   k_line_sweep_rp  (one-sided chain, lane per row)         where neither applies (factor beyond 4 GiB, EMG3D_TWIST=0);
    //   k_line_sweep_qc<..., BIG> (the same with 64-bit field offsets)      levels whose field arrays reach 4 GiB (q_big);
    //   k_line_sweep     (thread per line, 64-bit offsets)       such levels in the lexicographic order or with fewer lines, EMG3D_SWEEP=tpl.
    // EMG3D_QPL=<direction bit mask> (0: off), EMG3D_QPL_MAX_NL, EMG3D_QPL_FEW, EMG3D_QPL_M2 tune the first rule.
    // EMG3D_BATCH_TUNE=1 (default 0): with batched systems, choose between the scan kernel and the chain kernels by the
    // lines a LAUNCH carries (lines x systems) instead of the lines of one system -- the scan kernel does 4 x the
    // arithmetic and only pays while the chain kernels leave SIMDs idle.  8 systems at 128^3: 50.3 -> 46.4 ms per cycle.
    // Off by default because the kernel choice then depends on the batch size: a system's result agrees with its
    // stand-alone solve to rounding (1e-12) instead of bit for bit.
    int batch_tune = getenv("EMG3D_BATCH_TUNE") ? atoi(getenv("EMG3D_BATCH_TUNE")) : 0;
    // lexicographic order, lines of <= 16 blocks: hyperplane loop inside one workgroup instead of a launch per hyperplane
    int lex_loop = (int)LAB_ENV("EMG3D_LEX_LOOP", 1);
    // k_line_sweep_tha (smooth_tha.hpp: the two-sided solve in affine form, helper waves) on the mid levels the scan kernel served:
    // colour order, no split copies, lines of tha_min_nl .. tha_mid_nl (33..64) blocks, at least tha_min_lines lines per colour.
    // Returns the helper waves per half (0: another kernel serves).  Measured per launch (profiles/r04_rs_shapes.txt, r04_tha_ab.txt):
    // the launch is as long as its chain wave's work (~7 us + 0.6 us per step: 25 us at 64 blocks, 19 us at 40) whatever the
    // number of lines up to 2048 (one workgroup per CU at 8 lines each); the scan kernel grows with the lines (64-block lines:
    // 16 / 22 / 41 us at 512 / 1024 / 2048 lines per colour) and wins below ~1100; lines of <= 32 blocks stay with the scan kernel.
    // Batched handles take the same kernel: the choice must not depend on the batch size (a system stays bit for bit its own solve).
    int use_tha = (int)LAB_ENV("EMG3D_THA", 3);     // helper waves per half: 3 (lab: 2; 0: off, the scan kernel serves)
    i64 tha_min_nl = LAB_ENV("EMG3D_THA_MIN", 33), tha_mid_nl = LAB_ENV("EMG3D_THA_MID", 64), tha_min_lines_env = LAB_ENV("EMG3D_THA_MIN_LINES", 0);
    i64 tha_min_lines() const { return tha_min_lines_env > 0 ? tha_min_lines_env : (1100 * simds() + 1023) / 1024; }
    // It also serves lines of 65..128 blocks (tha_max_nl) when a colour has at most 2048 lines (tha_big_max_lines), i.e. ONE round
    // of workgroups at one per CU (142 KB of LDS at 128 blocks): 128 x 128 x 64, x- / y-lines: 74 -> 53 us per launch against
    // k_line_sweep_thm, the grid's F-cycle 6.71 -> 6.25 ms (profiles/r04_tha_long_lines.txt).  Level 0 of 128^3 (4032 lines per
    // colour = two rounds, each as long as its helper-bound forward pass) loses 103-105 to 86 us and keeps k_line_sweep_thm<8, ZS>
    // (lab: EMG3D_THA_BIG_LINES=8192; HISTORY R4.8).
    i64 tha_max_nl = LAB_ENV("EMG3D_THA_MAX", 128);
    i64 tha_big_max_lines_env = LAB_ENV("EMG3D_THA_BIG_LINES", 0);
    i64 tha_big_max_lines() const { return tha_big_max_lines_env > 0 ? tha_big_max_lines_env : (i64)THA_LPW * (simds() / 4); }
    int tha_split = (int)LAB_ENV("EMG3D_THA_SPLIT", 0);     // lab: also on mid levels that have split copies (EMG3D_SPLIT_MIN_CELLS)
    // LDS of a k_line_sweep_tha workgroup: up to 135 680 B dynamic (c128, 128-block lines) + THA_STATIC_LDS static.  Asked for once
    // per handle; a device or runtime that refuses it (or has less LDS per workgroup) gets the other kernels (tha_helpers -> 0)
    // instead of failing launches.
    mutable int tha_lds_state = -1;     // -1: not asked yet, 0: refused, 1: granted
    mutable i64 tha_lds_limit = 0;      // the device's LDS bytes per workgroup
    bool tha_lds_ok(i64 nL) const {
        if (tha_lds_state < 0) {
            int lim = 0;
            if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) { (void)hipGetLastError(); lim = 0; }
            tha_lds_limit = lim;
            const bool ok = lim >= THA_MAX_DYN_LDS + THA_STATIC_LDS && tha_attrs<T>(THA_MAX_DYN_LDS);
            tha_lds_state = ok ? 1 : 0;
        }
        return tha_lds_state == 1 && (i64)tha_lds_bytes<T, 3>((int)nL) <= (i64)THA_MAX_DYN_LDS &&
               (i64)tha_lds_bytes<T, 3>((int)nL) + THA_STATIC_LDS <= tha_lds_limit;
    }
    int tha_helpers(const Level<T>& L, int dir) const {
        if ((use_tha != 3 && use_tha != 2) || order != 1 || sweep_kernel != 0 || !use_twist || !rp_fits(L)) return 0;
        const i64 nL = L.nC[dir];
        if (nL < tha_min_nl || nL > tha_max_nl || !tha_lds_ok(nL)) return 0;
        const bool mid = nL >= tha_min_nl && nL <= tha_mid_nl && (!split_on(L) || tha_split);
        const bool big = nL > tha_mid_nl && nL <= tha_max_nl;
        if (!mid && !big) return 0;
        const int P = (dir == 0) ? 1 : 0, Q = (dir == 2) ? 1 : 2;
        const i64 lines = (L.nC[P] / 2) * (L.nC[Q] / 2);                // largest colour
        if (lines < tha_min_lines() || lines >= q_min_lines()) return 0;
        if (big && lines > tha_big_max_lines()) return 0;                 // (more than one round of workgroups at one per CU)
        return use_tha;
    }
    bool qpl(const Level<T>& L, int dir) const {
        if (!((use_qpl >> dir) & 1) || split_on(L) || sweep_kernel != 0 || tha_helpers(L, dir)) return false;
        const i64 cap = (L.nC[dir] >= qpl_m2_min) ? 256 : 128;     // 8 waves x 16 quads x M blocks per line
        const int P = (dir == 0) ? 1 : 0, Q = (dir == 2) ? 1 : 2;
        i64 lines = (L.nC[P] / 2) * (L.nC[Q] / 2);                  // per colour
        i64 max_nl = qpl_max_nl;
        if (batch_tune && nsys > 1 && order == 1) {                 // a launch carries nsys x the lines (see batch_tune)
            lines *= nsys;
            max_nl = std::max<i64>(4, qpl_max_nl / nsys);
        }
        const i64 maxnl = (order == 0 || lines <= qpl_few_lines()) ? cap : std::min<i64>(max_nl, cap);
        if (L.nC[dir] < qpl_min_nl || L.nC[dir] > maxnl || !rp_fits(L)) return false;
        return lines <= qpl_max_lines;
    }
    // (lines that fit ONE wave -- seg <= 16 -- get single-wave workgroups: fewer, fatter workgroups were measured slower, 8.54 -> 8.8 /
    // 9.1 / 10.5 ms per 128^3 F-cycle at 2 / 4 / 8 waves, HISTORY R5.13)
    // workgroup waves NW, blocks per quad M, quads per line seg (power of two, M * seg >= nL)
    void qpl_shape(i64 nL, int& NW, int& M, int& seg) const {
        M = (nL >= qpl_m2_min) ? 2 : 1;
        const i64 nch = (nL + M - 1) / M;
        seg = 4; while (seg < nch) seg *= 2;
        NW = seg <= 16 ? 1 : seg / 16;
    }
    // sweep = false: arguments for k_line_factor (un-split model arrays);
    // sweep = true : arguments for the sweep kernels (working copies).
    void line_args(Level<T>& L, int dir, LineArgs<T>& a, bool sweep) {
        if (dir == 0) { a.L = 0; a.P = 1; a.Q = 2; }
        else if (dir == 1) { a.L = 1; a.P = 0; a.Q = 2; }
        else { a.L = 2; a.P = 0; a.Q = 1; }
        const bool sp = sweep && split_on(L);
        // the split working copy of the x direction is always the transposed one
        const bool t = (sp && dir == 0) ? true : xt(L, dir);
        const int w = (dir == 0) ? 0 : 1;
        for (int q = 0; q < 3; ++q) {
            a.nC[q] = L.nC[q]; a.eta[q] = t ? L.etaT[q] : L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q];
        }
        a.fl = t ? L.flT : L.fl; a.cl = t ? L.clT : L.cl;
        a.split = sp ? 1 : 0;
        if (sp) { a.e = L.eW[w]; a.s = L.sW[w]; a.zeta = L.zetaW[w]; }
        else { a.e = t ? L.eT : L.e; a.s = t ? L.sT : L.s; a.zeta = t ? L.zetaT : L.zeta; }
        const i64 nP = L.nC[a.P], nQ = L.nC[a.Q];
        a.nA[0] = (nP - 0) / 2; a.nA[1] = (nP - 1) / 2;
        const i64 nB[2] = {(nQ - 0) / 2, (nQ - 1) / 2};
        a.nB2[0] = nB[0]; a.nB2[1] = nB[1];
        i64 o = 0;
        for (int c = 0; c < 4; ++c) { a.base[c] = o; o += a.nA[c & 1] * nB[c >> 1]; }
        a.nLinesTot = o;   // == (nP-1)*(nQ-1)
        a.fac = L.fac[dir];
        a.mid = L.fac[dir] ? L.fac_mid[dir] : L.nC[a.L] - 1;
        a.qm = (L.fac[dir] && L.fac_kind[dir] == 3) ? 2 : 0;       // 2: mirrored two-sided factor, k_line_sweep_thm
        a.fcomp = (L.fac[dir] && L.fac_kind[dir] == 4) ? 1 : 0;
        a.zsep = (sweep && L.zeta_sep && use_zsep) ? 1 : 0;
        a.sflag = (sweep && sflag_on(L, dir) && L.sflag[dir] && L.sflag_valid[dir]) ? L.sflag[dir] : nullptr;
        a.xcd = xcd_map;
        a.tile = q_tile;
        {
            const int ax[3] = {a.L, a.P, a.Q};
            a.rs.ihL = a.ih[a.L]; a.rs.ihP = a.ih[a.P]; a.rs.ihQ = a.ih[a.Q];
            a.rs.hL = a.h[a.L]; a.rs.hP = a.h[a.P]; a.rs.hQ = a.h[a.Q];
            a.rs.nL = (unsigned)a.nC[a.L]; a.rs.slot0 = 0;
            a.rs.nP = (unsigned)a.nC[a.P]; a.rs.nQ = (unsigned)a.nC[a.Q];
            a.rs.csL = (unsigned)a.cl.st[a.L]; a.rs.csP = (unsigned)a.cl.st[a.P]; a.rs.csQ = (unsigned)a.cl.st[a.Q];
            for (int c = 0; c < 3; ++c) {
                a.rs.off[c] = (unsigned)a.fl.off[ax[c]];
                for (int d = 0; d < 3; ++d) a.rs.st[c][d] = (unsigned)a.fl.st[ax[c]][ax[d]];
            }
        }
        a.qd = nullptr; a.qdn = 0;
        a.qpl = 0; a.qM = 0; a.seg = 0; a.qlpw = 0;
        if (qpl(L, dir)) { int NW, M, seg; qpl_shape(L.nC[a.L], NW, M, seg); a.qpl = NW; a.qM = M; a.seg = seg; }
        a.tha = tha_helpers(L, dir);
        a.mode = 0; a.cP = a.cQ = 0; a.cntA = a.cntB = 0; a.t = a.jQ0 = a.cnt = 0;
        a.bt = sweep ? batch(L) : Batch();
    }

    // Two-sided factorisation + k_line_sweep_thm for latency-bound launches: fewer than 8192 lines per colour (beyond that the
    // sweep is HBM bound and the quad-per-line kernel on the compact factor moves fewer bytes), strides within the 24-bit
    // multiplies of the kernel (twist_ok).
    // quad-per-line chain kernel for this (level, direction)?  Decided by the level's largest colour.
    bool q_on(const LineArgs<T>& a) const {
        return use_q >= 2 || (use_q == 1 && a.nA[0] * a.nB2[0] >= q_min_lines());
    }
    // Two-sided sweeps on the MIRRORED factorisation (k_line_sweep_thm: left blocks [l_i; T_i] upwards, right blocks
    // [l_j; T_{j-1}] downwards: the reference's accuracy; round 1's plain two-sided grouping was 1e-8 on ill-conditioned lines).
    bool thm_on(const Level<T>& L, const LineArgs<T>& a) const { return twist_ok(L, a); }
    bool twist_ok(const Level<T>& L, const LineArgs<T>& a) const {
        if (q_on(a)) return false;                    // the quad-per-line kernel has its own (compact, one-sided) factorisation
        if (!use_twist || !rp_fits(L) || L.nC[a.L] < 3) return false;
        const i64 nQ = L.nC[a.Q];
        const i64 maxlines = a.nA[0] * ((nQ - 0) / 2);
        if (maxlines >= twist_max_lines()) return false;
        const i64 lim24 = (i64)1 << 24;
        i64 mxs = 15 * a.nLinesTot * (i64)sizeof(T);
        for (int c = 0; c < 3; ++c) mxs = std::max(mxs, a.fl.st[c][a.L] * (i64)sizeof(T));
        mxs = std::max(mxs, a.cl.st[a.L] * 8);
        // the two-sided kernels form the factor offset block * stride + entry in 32 bits: the whole factor of
        // the direction must stay below 4 GiB (160 x 160 x 768 complex would wrap silently otherwise)
        const i64 fac_bytes = 15 * a.nLinesTot * L.nC[a.L] * (i64)sizeof(T);
        return mxs < lim24 && L.nC[a.L] < lim24 && fac_bytes < ((i64)1 << 32);
    }

    // Layout of the cached factorisation of (level, direction), i.e. which kernel family will sweep it (Level::fac_kind): 4 = compact
    // (G and r: 11 numbers per block) wherever the quad-per-line kernel serves (smooth_qc.hpp), 3 = mirrored two-sided
    // (k_line_sweep_thm / _tha), 0 = one-sided, 15 numbers per block (scan kernel, k_line_sweep_rp, thread per line).  a: line_args(L, dir, a, false).
    int factor_kind(const Level<T>& L, const LineArgs<T>& a) const {
        if (!a.qpl && thm_on(L, a)) return 3;
        return (!a.qpl && (rp_fits(L) || q_big(L)) && q_on(a) && sweep_kernel == 0) ? 4 : 0;
    }
    // The kernel instantiation the colour launches of (level, direction) select, by name as `rocprofv3 --kernel-trace` and
    // emg3d_mg_last_sweep_kernel show it, from the level's SHAPE alone -- the same predicates ensure_factor / launch_sweep use, no
    // device memory, no launch (emg3d_sweep_plan: host-side tests of the selection on devices of other sizes).  info: [0] lines of
    // the largest colour, [1] lines per wave (qc, rp: per wave; thm: per pair of waves; tha: per workgroup; qpl: lines per workgroup),
    // [2] rounds of waves / workgroups of the largest colour's launch, [3] factor layout (factor_kind), [4] sweeps run on
    // parity-split working copies, [5] 64-bit field offsets.
    void plan_sweep(Level<T>& L, int dir, char* name, i64 info[6]) {
        LineArgs<T> a;
        line_args(L, dir, a, false);
        const int kind = factor_kind(L, a);
        const bool big = q_big(L) && !rp_fits(L), rp = rp_fits(L) || big;
        const i64 nmax = a.nA[0] * a.nB2[0], S = simds();
        const char* tn = sizeof(T) == 16 ? "c128" : "f64";
        i64 lpw = 0, rounds = 0;
        if (kind == 3 && a.tha) {
            snprintf(name, 64, "k_line_sweep_tha<%s,%d>", tn, a.tha);
            lpw = THA_LPW; rounds = (((nmax + THA_LPW - 1) / THA_LPW) * nsys + S / 4 - 1) / (S / 4);
        } else if (kind == 3) {
            lpw = th_lines_per_pair(a);
            snprintf(name, 64, "k_line_sweep_thm<%s,%d,%d>", tn, tw_stages ? tw_stages : 3, (int)lpw);
            rounds = (2 * ((nmax + lpw - 1) / lpw) * nsys + S - 1) / S;
        } else if (a.qpl) {
            snprintf(name, 64, "%s<%s,%d,%d>", (a.qpl == 1 && a.qM == 1 && order == 1 && a.seg <= qpl_chain_seg) ? "k_line_sweep_qpl_chain" : "k_line_sweep_qpl",
                     tn, a.qpl, a.qM);
            lpw = (16 * a.qpl) / a.seg;
            rounds = (((nmax + lpw - 1) / lpw) * a.qpl * nsys + S - 1) / S;
        } else if (rp && kind == 4) {
            const int inst = big ? 16 : q_lpw ? q_lpw : (nmax >= q_min_lines() ? 16 : 4);
            snprintf(name, 64, "%s<%s,%d,%d>", big ? "k_line_sweep_qc_big" : "k_line_sweep_qc", tn, q_stages_for(inst, nmax), inst);
            lpw = inst == 16 ? q_balanced_lpw(nmax * nsys) : inst;
            rounds = (((nmax + lpw - 1) / lpw) * nsys + S - 1) / S;
        } else if (rp) {
            lpw = force_lpw ? force_lpw : (nmax >= 8 * S ? 8 : 4);
            snprintf(name, 64, "k_line_sweep_rp<%s,%d>", tn, (lpw == 8 || lpw == 12) ? (int)lpw : 4);
            rounds = (((nmax + lpw - 1) / lpw) * nsys + S - 1) / S;
        } else {
            snprintf(name, 64, "k_line_sweep<%s>", tn);
            lpw = 64; rounds = (((nmax + 63) / 64) * nsys + S - 1) / S;
        }
        info[0] = nmax; info[1] = lpw; info[2] = rounds; info[3] = kind; info[4] = split_on(L) ? 1 : 0; info[5] = big ? 1 : 0;
    }
    void ensure_factor(Level<T>& L, int dir) {
        if (xt(L, dir)) ensure_transposed_model(L);
        if (L.fac[dir]) return;
        LineArgs<T> a;
        line_args(L, dir, a, false);
        const i64 per_line = a.qpl ? (i64)a.qM * a.seg : L.nC[a.L];
        const int kind = factor_kind(L, a);
        L.fac[dir] = dalloc<T>(a.nLinesTot * per_line * (kind == 4 ? 11 : 15));
        L.fac_lines[dir] = a.nLinesTot;
        L.fac_mid[dir] = L.nC[a.L] - 1;     // one-sided, unless ...
        L.fac_kind[dir] = kind;
        if (kind == 3) {                    // ... the mirrored two-sided factorisation serves
            L.fac_mid[dir] = qm_mid(L.nC[a.L]);
            thm_attrs();
        }
        compute_factor(L, dir);
    }
    // (re)compute the cached factorisation of (level, direction) into its buffer: the model may have changed (set_smu0)
    void compute_factor(Level<T>& L, int dir) {
        LineArgs<T> a;
        line_args(L, dir, a, false);
        a.fac = L.fac[dir];
        a.fcomp = L.fac_kind[dir] == 4;
        if (L.fac_kind[dir] == 3) {  // all four colours in one launch
            a.mid = L.fac_mid[dir];
            const i64 nQ_ = L.nC[a.Q];
            const i64 nmax_ = a.nA[0] * ((nQ_ - 0) / 2);
            a.mode = 3;
            if (nmax_ > 0)
                MG_LAUNCH(k_line_factor_m<T>, dim3((unsigned)((nmax_ + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK), 4),
                                   dim3(EMG_LINE_BLOCK), 0, stream, a);
            check_launch();
            return;
        }
        a.mid = L.fac_mid[dir];
        const i64 nQ = L.nC[a.Q];
        const i64 nB[2] = {(nQ - 0) / 2, (nQ - 1) / 2};
        // the four colours are independent here: one launch (blockIdx.y = colour), 4 x the threads of a
        // chain-latency-bound kernel (128^3 level 0: 4 x 0.45 ms -> one launch)
        a.mode = 3; a.cP = a.cQ = 0; a.cntA = a.cntB = 0;
        const i64 nmax = a.nA[0] * nB[0];
#ifdef EMG3D_LAB
        // lab: one launch per colour -- the thread-per-line factor recurrence with only ONE colour's lines on the chip is
        // the lower bound of a sweep that recomputes the factor instead of reading it (profiles/HISTORY.md, recompute)
        if (LAB_ENV("EMG3D_FACTOR_PER_COLOUR", 0)) {
            for (int c = 0; c < 4; ++c) {
                a.mode = 0; a.cP = c & 1; a.cQ = c >> 1; a.cntA = a.nA[a.cP]; a.cntB = nB[a.cQ];
                const i64 n = a.cntA * a.cntB;
                if (n > 0)
                    MG_LAUNCH(k_line_factor<T>, dim3((unsigned)((n + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK)),
                                       dim3(EMG_LINE_BLOCK), 0, stream, a);
            }
            check_launch();
            return;
        }
#endif
        if (nmax > 0)
            MG_LAUNCH(k_line_factor<T>, dim3((unsigned)((nmax + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK), 4),
                               dim3(EMG_LINE_BLOCK), 0, stream, a);
        check_launch();
    }

    // n independent lines: row-parallel kernel (8 lanes per line) by default;
    // EMG3D_SWEEP=tpl selects the thread-per-line kernel (A/B + debugging).
    // Lines per wave: with few lines the recurrence is latency bound and more
    // waves win (4 lines/wave); with many lines the sweep is HBM bound and
    // fewer, fuller waves move fewer bytes (8 lines/wave).  Measured on MI355X:
    // 128^3 (4032 lines/colour) 0.67 vs 0.75 ms, 256^3 (16129) 5.6 vs 4.6 ms.
    // workgroups of a row-parallel launch (a multiple of the 8 XCDs when the XCD-aware map is on:
    // the kernels drop the workgroups past the last line)
    unsigned rp_grid(i64 nt) const {
        const i64 nb = (nt + EMG_RP_BLOCK - 1) / EMG_RP_BLOCK;
        return (unsigned)(xcd_map ? ((nb + 7) / 8) * 8 : nb);
    }
    template <int LPW>
    void launch_rp(const LineArgs<T>& a, i64 n) {
        const i64 nwaves = (n + LPW - 1) / LPW;
        const i64 nt = nwaves * 64;
        MG_LAUNCH((k_line_sweep_rp<T, LPW>), bgrid(rp_grid(nt)), dim3(EMG_RP_BLOCK), 0, stream, a);
    }
    void launch_qpl(const LineArgs<T>& a, i64 n) {
        const int NW = a.qpl;
        const i64 lpg = (16 * NW) / a.seg;              // lines per workgroup
        const i64 nb = (n + lpg - 1) / lpg;
        if (broken) return;
        // (single-wave workgroups in colour order: with the descriptors written when the factor was built, ensure_qdesc)
        const bool chain = NW == 1 && a.qM == 1 && a.mode == 0 && a.seg <= qpl_chain_seg;
        qpl_launch<T>(NW, a.qM, false, (NW == 1 && a.qd) ? 2 : 0, chain, bgrid(qpl_grid(nb)), stream, a);
        if (chain) note_kernel("k_line_sweep_qpl_chain", NW, a.qM);
    }
    unsigned qpl_grid(i64 nb) const { return (unsigned)(xcd_map ? ((nb + 7) / 8) * 8 : nb); }
    // Descriptors of the scan kernel's colour launches on levels of short lines (one wave per workgroup): everything of the
    // prologue that depends on grid and model only -- 11 element offsets / flags and 15 coefficient products per lane and block --
    // is computed ONCE, by the kernel's own prologue in generating mode, and loaded by the 7 launches of every smoothing call
    // thereafter (HISTORY R5.12).  176 B per thread: only where a colour launch has at most qdesc_max_threads threads and the factor
    // has fewer than 2^32 entries.  Measured (profiles/r05_qdesc_ab.txt, 128^3 F-cycle, three alternating repetitions): off 8.57 /
    // 8.54 / 8.50 ms; launches of <= 9000 threads (308 of the 420 on <= 16-block lines) 8.456 / 8.458 / 8.451; <= 40 000 threads (all
    // 420) 8.495 / 8.468 / 8.48; the 32-block level too (two blocks per quad, 65 k threads) 8.72: beyond ~8 k threads the table costs
    // more to read than the arithmetic it replaces.  In-kernel stamps at 128 x 4 x 4: 8330 -> 7500 cycles.
    int use_qdesc = (int)LAB_ENV("EMG3D_QDESC", 1);
    i64 qdesc_max_threads_env = LAB_ENV("EMG3D_QDESC_MAX", 0);
    i64 qdesc_max_threads() const { return qdesc_max_threads_env > 0 ? qdesc_max_threads_env : (9000 * simds() + 1023) / 1024; }
    void ensure_qdesc(Level<T>& L, int dir) {
        if (!use_qdesc || order != 1 || L.qd[dir][0] || L.qdn[dir][0] == ~0u) return;
        LineArgs<T> a;
        line_args(L, dir, a, true);
        L.qdn[dir][0] = ~0u;                                    // (asked once)
        if (a.qpl != 1 || !L.fac[dir]) return;
        const i64 nQ = L.nC[a.Q];
        const i64 nB[2] = {(nQ - 0) / 2, (nQ - 1) / 2};
        const i64 lpg = 16 / a.seg, per = (i64)a.qM * a.seg;
        if (a.nLinesTot * 15 * per >= ((i64)1 << 32)) return;
        if (((a.nA[0] * nB[0] + lpg - 1) / lpg) * 64 > qdesc_max_threads()) return;
        a.bt = Batch();                                         // (the descriptors do not depend on the system)
        for (int c = 0; c < 4; ++c) {
            a.mode = 0; a.cP = c & 1; a.cQ = c >> 1;
            a.cntA = a.nA[a.cP]; a.cntB = nB[a.cQ];
            a.rs.slot0 = (unsigned)a.base[c];
            const i64 n = a.cntA * a.cntB;
            if (n <= 0) continue;
            const unsigned grid = qpl_grid((n + lpg - 1) / lpg);
            const i64 nthreads = (i64)grid * 64;
            void* tab = try_alloc<char>(nthreads * a.qM * (3 * 16 + 8 * 16));     // (pure optimisation data: without it the kernel computes)
            if (!tab) return;
            a.qd = tab; a.qdn = (unsigned)nthreads;
            if (!broken) qpl_launch<T>(1, a.qM, false, 1, false, dim3(grid), stream, a);
            L.qd[dir][c] = tab; L.qdn[dir][c] = (unsigned)nthreads;
        }
        check_launch();
    }
    // name of the kernel instantiation the last line-sweep launch selected (bench.py's roofline object and the
    // sweep-level parity tests report it instead of guessing from the grid size)
    char sweep_name[64] = "";
    char res_name[64] = "";            // the same for the most recent residual launch
    void note_kernel(const char* base, int p1, int p2) {
        const char* tn = sizeof(T) == 16 ? "c128" : "f64";
        if (p2 >= 0) snprintf(sweep_name, sizeof sweep_name, "%s<%s,%d,%d>", base, tn, p1, p2);
        else if (p1 >= 0) snprintf(sweep_name, sizeof sweep_name, "%s<%s,%d>", base, tn, p1);
        else snprintf(sweep_name, sizeof sweep_name, "%s<%s>", base, tn);
    }
    // A launch of the quad kernel at 16 lines per wave is ONE wave per SIMD (three prefetch stages: 322+ registers; with two stages a
    // second wave fits, but one full wave per SIMD is the faster form -- 256^3: 9 lines per wave on two waves per SIMD 0.90 against
    // 0.73 ms).  Its waves all last the same time, so a launch of W waves on C = SIMDs wave slots lasts ceil(W / C) rounds: 448^3 --
    // 3136 waves = 3.06 rounds of 1024 -- pays four (12.9 % of the roofline where 512^3, exactly four rounds, reaches 16 %).  Deal the
    // lines evenly instead: the fewest rounds r that 16 lines per wave allow, then ceil(lines / (C r)) lines per wave (>= 8: levels of
    // 8192 ... 16383 lines per colour take the same path, launch_sweep).  Measured by size (profiles/r05_balanced_lpw.txt, dense source, % of the
    // algorithmic roofline): 288^3 11.6 -> 13.9, 320^3 14.8 -> 15.8, 368^3 12.5 -> 14.9, 384^3 13.4 -> 15.1, 448^3 12.9 -> 14.4, 480^3
    // 14.1 -> 14.5; 256^3, 352^3, 512^3 (whole rounds already) unchanged.  Bit-identical (a line's arithmetic does not know its
    // wave).
    int q_balanced_lpw(i64 lines) const {
        const i64 cap = simd_count();
        const i64 rounds = std::max<i64>(1, (lines + 16 * cap - 1) / (16 * cap));
        const i64 lpw = (lines + cap * rounds - 1) / (cap * rounds);
        return (int)std::min<i64>(16, std::max<i64>(lpw, 8));
    }
    // lpw: the instantiation (16 | 8 | 4 | 2 lines per wave; big: the 16-line one with 64-bit field offsets); the 16-line
    // instantiation runs at the balanced number of lines per wave
    void launch_qc(const LineArgs<T>& a0, i64 n, int lpw, bool big) {
        LineArgs<T> a = a0;
        if (big) lpw = 16;
        a.qlpw = (lpw == 16) ? q_balanced_lpw(n * nsys) : lpw;
        const i64 nt = ((n + a.qlpw - 1) / a.qlpw) * 64;
        if (!broken) qc_launch<T>(q_stages_for(lpw, a.nA[0] * a.nB2[0]), lpw, a.zsep != 0, big, bgrid(rp_grid(nt)), stream, a);
    }
    // lab: k_line_sweep_thm can keep the last KL forward steps of a half in LDS (smooth_thm.hpp; KL by lines per pair of waves
    // so that the workgroup stays within the CU's 160 KB).  Measured at 128^3: counted traffic 491 -> 453 MB per launch,
    // launch 102.3 -> 103.7 us (profiles/HISTORY.md) -- the saving sits in steps during which every wave of the launch is
    // off the memory system at the same time.  Off; EMG3D_THM_LIFO=1 switches it on in the lab build.
    int thm_lifo = (int)LAB_ENV("EMG3D_THM_LIFO", 0);
    // More than 64 KB of LDS per workgroup must be asked for, per kernel instantiation and device; done when the factor of
    // a two-sided level is built, i.e. before the launches are captured into a graph.  (Only the lab build's LIFO variants need
    // it; k_line_sweep_tha's dynamic LDS is asked for where the kernel is selected: tha_lds_ok.)
    void thm_attrs() {
#ifdef EMG3D_LAB
        static bool done[64] = {false};
        if (device < 0 || device >= 64 || done[device]) return;
        done[device] = true;
        thm_lifo_attrs<T>();
#endif
    }
    void launch_thm_l(const LineArgs<T>& a, i64 n, int LPW) {
        const i64 npairs = (n + LPW - 1) / LPW;
        const i64 nb = (npairs * 128 + EMG_RP_BLOCK - 1) / EMG_RP_BLOCK;
        const unsigned grid = (unsigned)(xcd_map ? ((nb + 7) / 8) * 8 : nb);
        const int stages = tw_stages ? tw_stages : 3;
        note_kernel("k_line_sweep_thm", stages, LPW);
        if (!broken) thm_launch<T>(stages, LPW, thm_lifo != 0, a.zsep != 0, bgrid(grid), stream, a);
    }
    void launch_tha(const LineArgs<T>& a, i64 n, int NH) {
        const i64 nb = (n + THA_LPW - 1) / THA_LPW;
        const unsigned grid = (unsigned)(xcd_map ? ((nb + 7) / 8) * 8 : nb);
        snprintf(sweep_name, sizeof sweep_name, "k_line_sweep_tha<%s,%d>", sizeof(T) == 16 ? "c128" : "f64", NH);
        const size_t dyn = NH == 2 ? tha_lds_bytes<T, 2>((int)a.nC[a.L]) : tha_lds_bytes<T, 3>((int)a.nC[a.L]);
        if (!broken) tha_launch<T>(NH, a.zsep != 0, bgrid(grid), dyn, stream, a);
    }
    void launch_thm(const LineArgs<T>& a, i64 n) {
        if (a.tha) {                    // mid levels: the affine kernel with three helper waves per half (HISTORY R4.6-R4.7)
#ifdef EMG3D_LAB
            if (a.tha == 2) { launch_tha(a, n, 2); return; }
#endif
            launch_tha(a, n, 3);
            return;
        }
        const int lpw = th_lines_per_pair(a);
        launch_thm_l(a, n, (lpw == 4 || lpw == 12) ? lpw : 8);
    }
    void launch_sweep(const LineArgs<T>& a, i64 n, bool rp, bool big = false) {
        if (log_launches) fprintf(stderr, "[sweep] nC %lld %lld %lld L %d lines %lld kernel %s split %d\n", (long long)a.nC[0], (long long)a.nC[1], (long long)a.nC[2], a.L, (long long)n,
                                  a.qm == 2 ? "thm" : a.qpl ? "qpl" : (rp && a.fcomp) ? "qc" : rp ? "rp" : "tpl", a.split);
        if (a.qm == 2) {
            launch_thm(a, n);
        } else if (a.qpl) {
            note_kernel("k_line_sweep_qpl", a.qpl, a.qM);
            launch_qpl(a, n);
        } else if (rp && a.fcomp) {
            // lines per wave by the level's largest colour: aim at >= ~1000 waves (one per SIMD) before filling lanes
            const i64 nmax = a.nA[0] * a.nB2[0];
            // (8192 ... 16383 lines: the 16-line instantiation at ceil(lines / SIMDs) = 8 ... 16 lines per wave -- ONE round of waves --
            // instead of 8 lines per wave in up to two, q_balanced_lpw)
            const int lpw = big ? 16 : q_lpw ? q_lpw : (nmax >= q_min_lines() ? 16 : 4);
            note_kernel(big ? "k_line_sweep_qc_big" : "k_line_sweep_qc", q_stages_for(lpw, nmax), lpw);
            launch_qc(a, n, lpw, big);
        } else if (rp) {
            // by the level's largest colour, not by this colour's own count: the colours of one level
            // must not straddle the threshold (256 x 128 x 128: 8192 / 8128 / 8064 / 8001 lines; 8 lines per
            // wave 0.20 ms per launch, 4 lines per wave 0.30 ms)
            const int lpw = force_lpw ? force_lpw : (a.nA[0] * a.nB2[0] >= 8 * simds() ? 8 : 4);
            note_kernel("k_line_sweep_rp", (lpw == 8 || lpw == 12) ? lpw : 4, -1);
            if (lpw == 8) launch_rp<8>(a, n);
            else if (lpw == 12) launch_rp<12>(a, n);
            else launch_rp<4>(a, n);
        } else {
            note_kernel("k_line_sweep", -1, -1);
            MG_LAUNCH(k_line_sweep<T>, bgrid_y((unsigned)((n + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK)),
                               dim3(EMG_LINE_BLOCK), 0, stream, a);
        }
    }

    // Bring e (and s, if stale) into the working copy used by direction `dir`,
    // or write the smoothed e back.
    // allocation-only counterpart of to_work (used by the dry run before graph capture)
    void prepare_work(Level<T>& L, int dir) {
        ensure_sflags(L, dir);          // (dry: allocation only)
        if (home_on(L) || home_lvl(L)) ensure_work(L, 1);
        if (split_on(L)) ensure_work(L, (dir == 0) ? 0 : 1);
        else if (xt(L, dir)) ensure_transposed_model(L);
    }
    void to_work(Level<T>& L, int dir) {
        ensure_sflags(L, dir);
        if (split_on(L)) {
            const int w = (dir == 0) ? 0 : 1;
            ensure_work(L, w);
            if (home_lvl(L)) {          // field and source live in the x-split copies
                if (w == 0) {
                    if (!L.sW_valid[0]) { convert_field(L, L.sW[0], L.sW[1], 2, true); L.sW_valid[0] = true; }
                    convert_field(L, L.eW[0], L.eW[1], 2, true);
                }
                return;
            }
            if (!L.sW_valid[w]) { convert_field(L, L.sW[w], L.s, w, true); L.sW_valid[w] = true; }
            if (!home_on(L)) convert_field(L, L.eW[w], L.e, w, true);
            else {      // every system's field into the x-split copy first (frozen systems keep theirs there)
                e_to_w1(L);
                if (w == 0) convert_field(L, L.eW[0], L.eW[1], 2, true);
            }
        } else if (xt(L, dir)) {
            if (!L.sT_valid) { convert_field(L, L.sT, L.s, -1, true); L.sT_valid = true; }
            convert_field(L, L.eT, L.e, -1, true);
        }
    }
    void from_work(Level<T>& L, int dir) {
        if (home_on(L) || home_lvl(L)) {       // the field stays in (x-lines: goes to) the x-split copy
            if (dir == 0) convert_field(L, L.eW[1], L.eW[0], 2, false);
        } else if (split_on(L)) convert_field(L, L.e, L.eW[(dir == 0) ? 0 : 1], (dir == 0) ? 0 : 1, false);
        else if (xt(L, dir)) convert_field(L, L.e, L.eT, -1, false);
    }
    int work_id(Level<T>& L, int dir) { return split_on(L) ? ((dir == 0) ? 0 : 1) : (xt(L, dir) ? 2 : 3 + dir); }

    // ---- placement of level 0's WRITTEN working copies (HISTORY R5.18, R6.1) -----------------------------------------------
    // A level-0 colour launch at 256^3 lasts 0.62 ... 0.73 ms depending on which piece of physical memory the working copy it
    // WRITES (eW[w]) got -- candidates fall into two classes 6-9 % apart, which one a hipMalloc returns differs from box to box and
    // from allocation to allocation; the blocks it only reads (factor, source) matter < 1 % (profiles/r06_placement.txt) -- and
    // hipMalloc does not expose it.  So the handle tries: right before the first launch sequence that sweeps on eW[w] is captured
    // (no graph holds the pointer yet) up to place_tries candidate blocks -- the one it has, then fresh ones, all held until the end so
    // that every candidate is another piece of memory -- are timed with one sweep each (4 ms per candidate at 256^3) until
    // one of the fast class is in hand; it stays, the others go back to the pool / driver.  The kept block is parked under its role
    // when the handle goes (DevicePool tags): later handles of the process take it without searching.  Only level 0, only working
    // copies of >= place_min_bytes (256 MiB); a candidate that cannot be allocated ends the search quietly.  The values a sweep
    // computes do not depend on the block it runs on (cycles bit-identical).  EMG3D_PLACE_TRIES=<n> (default 12; 0: off).
    int place_tries = getenv("EMG3D_PLACE_TRIES") ? atoi(getenv("EMG3D_PLACE_TRIES")) : 12;
    i64 place_min_bytes = LAB_ENV("EMG3D_PLACE_MIN_MB", 256) << 20;
    int place_reps = (int)LAB_ENV("EMG3D_PLACE_REPS", 1);
    double place_gap = 1.075;
    static const int PLACE_MAX = 16;
    struct PlaceRec { int tries = 0, kept = 0, reused = 0; float ms[PLACE_MAX] = {0}; };
    PlaceRec place_rec[2];
    bool placed[2] = {false, false};
    static bool lr_has(int lr, int dir) {
        return dir == 0 ? (lr == 1 || lr == 5 || lr == 6 || lr == 7) : dir == 1 ? (lr == 2 || lr == 4 || lr == 6 || lr == 7)
                                                                              : (lr == 3 || lr == 4 || lr == 5 || lr == 7);
    }
    bool place_applies(const Level<T>& L) const {
        return place_tries > 1 && order == 1 && sweep_kernel == 0 && !trace && split_on(L) &&
               (i64)nsys * L.nE * (i64)sizeof(T) >= place_min_bytes;
    }
    bool placement_pending(int lr_dir) const {
        const Level<T>& L = *lv0;
        if (!place_applies(L)) return false;
        const int lr = current_lr_dir(lr_dir, L.nC);
        return (lr_has(lr, 0) && !placed[0]) || ((lr_has(lr, 1) || lr_has(lr, 2)) && !placed[1]);
    }
    bool alloc_quiet = false;       // raw_alloc: a failing hipMalloc is the caller's business (no message, err untouched)
    template <class U>
    U* try_alloc(i64 n) {
        const int keep = err;
        const i64 keep_bytes = bytes;
        alloc_quiet = true;
        U* p = dalloc<U>(n);
        alloc_quiet = false;
        if (!p) { err = keep; bytes = keep_bytes; }
        return p;
    }
    float time_sweeps(Level<T>& L, int dir, int reps) {
        hipEvent_t t0 = nullptr, t1 = nullptr;
        float ms = -1.f;
        if (hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess && hipEventRecord(t0, stream) == hipSuccess) {
            for (int i = 0; i < reps; ++i) smooth_line(L, dir, 1, false, false);
            if (hipEventRecord(t1, stream) == hipSuccess && hipEventSynchronize(t1) == hipSuccess &&
                hipEventElapsedTime(&ms, t0, t1) == hipSuccess) ms /= (float)reps;
            else ms = -1.f;
        }
        if (ms < 0.f) (void)hipGetLastError();
        if (t0) hipEventDestroy(t0);
        if (t1) hipEventDestroy(t1);
        return ms;
    }
    void place_level0(int lr_dir) {
        Level<T>& L = *lv0;
        if (dry || !placement_pending(lr_dir)) return;
        const int lr = current_lr_dir(lr_dir, L.nC);
        bool moved = false;
        for (int w = 0; w < 2; ++w) {
            const int dir = (w == 0) ? (lr_has(lr, 0) ? 0 : -1) : lr_has(lr, 1) ? 1 : lr_has(lr, 2) ? 2 : -1;
            if (placed[w] || dir < 0 || !L.eW[w] || !L.sW[w] || !L.fac[dir]) continue;
            placed[w] = true;
            if (w == 1) e_to_ref(L);            // (the field may be at home in eW[1]: back to the reference layout first)
            if (!L.sW_valid[w]) { convert_field(L, L.sW[w], L.s, w, true); L.sW_valid[w] = true; }
            ensure_sflags(L, dir);
            const i64 n = (i64)nsys * L.nE;
            PlaceRec& R = place_rec[w];
            R = PlaceRec();
            T* cands[PLACE_MAX];
            cands[0] = L.eW[w];
            int best = 0;
            float worst = 0.f;
            const int nt = std::min<int>(place_tries, PLACE_MAX);
            for (int k = 0; k < nt; ++k) {
                T* c = (k == 0) ? cands[0] : try_alloc<T>(n);
                if (!c) break;
                cands[k] = c;
                hipMemsetAsync(c, 0, (size_t)n * sizeof(T), stream);
                L.eW[w] = c;
                if (k == 0) (void)time_sweeps(L, dir, 1);       // (the first launches of a kernel in a process: code upload)
                const float ms = time_sweeps(L, dir, place_reps);
                R.ms[k] = ms;
                R.tries = k + 1;
                if (ms < 0.f) break;
                if (ms < R.ms[best]) best = k;
                worst = std::max(worst, ms);
                // the candidates fall into classes (about 2.60 / 2.46 / 2.28 ms per 256^3 sweep): holding one that is place_gap faster than
                // the slowest seen -- fast against mid, or fast against slow -- ends the search; mid against slow (5 %) does not
                if ((double)R.ms[best] * place_gap < (double)worst) break;
            }
            if (R.ms[best] < 0.f) best = 0;
            R.kept = best;
            L.eW[w] = cands[best];
            block_tag[cands[best]] = 1 + w;
            for (int k = 0; k < R.tries; ++k) if (k != best) release(cands[k]);
            moved |= best != 0;
            if (getenv("EMG3D_LOG_SETUP")) {
                fprintf(stderr, "[place] working copy %d (%s-lines, %.0f MB): kept candidate %d of %d;", w, dir == 0 ? "x" : dir == 1 ? "y" : "z",
                        (double)n * sizeof(T) / 1048576.0, best, R.tries);
                for (int k = 0; k < R.tries; ++k) fprintf(stderr, " %.3f", R.ms[k]);
                fprintf(stderr, " ms per sweep\n");
            }
        }
        if (moved) drop_graphs();       // (none holds these pointers on the usual paths; a graph of an x-only cycle may)
        check_launch();
    }

    // nu sweeps along `dir`; conv_in / conv_out: convert e to / from the working copy
    void smooth_line(Level<T>& L, int dir, int nu, bool conv_in = true, bool conv_out = true) {
        if (nu <= 0) return;
        ensure_factor(L, dir);
        ensure_qdesc(L, dir);
        if (dry) { prepare_work(L, dir); return; }
        if (conv_in) to_work(L, dir);
        ensure_sflags(L, dir);          // (valid already inside a captured sequence: refresh_level0_source)
        LineArgs<T> a;
        line_args(L, dir, a, true);
        const bool big = q_big(L) && !rp_fits(L);          // 64-bit field offsets (a.fcomp is set: ensure_factor)
        const bool rp = rp_fits(L) || big;
        const i64 nP = L.nC[a.P], nQ = L.nC[a.Q];
        const i64 nB[2] = {(nQ - 0) / 2, (nQ - 1) / 2};
        int iback = 0;
        int last_c = -1;
        for (int it = 0; it < nu; ++it) {
            iback = 1 - iback;   // first sweep runs backward (core.py:552, 569)
            if (order == 1) {
                for (int ch = 0; ch < 4; ++ch) {
                    const int c = iback ? colour_perm_b[ch] : colour_perm[ch];
                    // A line update is a projection: re-solving a colour whose
                    // neighbours (all of other colours) have not changed since
                    // its last update reproduces the same values.  A backward sweep
                    // ends with the colour the next forward sweep starts with: the
                    // repeated colour at each turn-around is skipped (identical
                    // result up to rounding).
                    if (skip_idempotent && c == last_c) continue;
                    last_c = c;
                    a.mode = 0; a.cP = c & 1; a.cQ = c >> 1;
                    a.cntA = a.nA[a.cP]; a.cntB = nB[a.cQ];
                    a.rs.slot0 = (unsigned)a.base[c];
                    a.qd = L.qd[dir][c]; a.qdn = (L.qd[dir][c] ? L.qdn[dir][c] : 0u);
                    const i64 n = a.cntA * a.cntB;
                    if (n <= 0) continue;
                    launch_sweep(a, n, rp, big);
                }
            } else {
                const i64 tmin = 3, tmax = (nP - 1) + 2 * (nQ - 1);
                if (lex_loop && a.qpl && a.qM == 1 && a.seg <= 16) {
                    // short lines: ONE workgroup per system loops over the hyperplanes (k_line_sweep_qpl mode 2) --
                    // the same line solves in the same order as the launches below, without the launch boundaries
                    LineArgs<T> b = a;
                    b.mode = 2; b.t = tmin; b.jQ0 = tmax; b.cnt = iback; b.xcd = 0;
                    const i64 maxn = std::min<i64>(nQ - 1, (nP - 1) / 2 + 1), quads = maxn * a.seg;
                    const int lnw = quads <= 16 ? 1 : quads <= 32 ? 2 : quads <= 64 ? 4 : 8;
                    if (!broken) qpl_launch<T>(lnw, 1, true, 0, false, bgrid(1), stream, b);
                    note_kernel("k_line_sweep_qpl", lnw, 1);
                    continue;
                }
                for (i64 th = tmin; th <= tmax; ++th) {
                    const i64 tt = iback ? tmax - (th - tmin) : th;
                    i64 lo = tt - (nP - 1);                 // jQ >= ceil(lo/2)
                    i64 jQ0 = lo <= 0 ? 1 : (lo + 1) / 2;
                    if (jQ0 < 1) jQ0 = 1;
                    i64 jQ1 = (tt - 1) / 2;
                    if (jQ1 > nQ - 1) jQ1 = nQ - 1;
                    const i64 n = jQ1 - jQ0 + 1;
                    if (n <= 0) continue;
                    a.mode = 1; a.t = tt; a.jQ0 = jQ0; a.cnt = n;
                    launch_sweep(a, n, rp);
                }
            }
        }
        if (conv_out) from_work(L, dir);
        check_launch();
    }

    void smooth_point(Level<T>& L, int nu) {
        if (dry) return;
        if (&L == lv0.get()) e_to_ref(L);
        const bool hl = home_lvl(L) && L.e_home == 1;        // point smoothing of a level that lives in the x-split copies: through e, s
        if (hl) { convert_field(L, L.e, L.eW[1], 1, false); convert_field(L, L.s, L.sW[1], 1, false); }
        PointArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = L.nC[q]; a.eta[q] = L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q]; }
        a.fl = L.fl; a.e = L.e; a.s = L.s; a.zeta = L.zeta; a.bt = batch(L);
        a.col = 0; a.t = 0; a.cnt[0] = a.cnt[1] = a.cnt[2] = 0;
        int iback = 0;
        int last_c = -1;
        for (int it = 0; it < nu; ++it) {
            iback = 1 - iback;
            if (order == 1) {
                for (int ch = 0; ch < 8; ++ch) {
                    const int c = iback ? point_perm_b[ch] : point_perm[ch];
                    if (skip_idempotent && c == last_c) continue;   // see smooth_line
                    last_c = c;
                    a.mode = 0; a.col = c;
                    for (int q = 0; q < 3; ++q) a.cnt[q] = (L.nC[q] - ((c >> q) & 1)) / 2;
                    const i64 n = a.cnt[0] * a.cnt[1] * a.cnt[2];
                    if (n <= 0) continue;
                    MG_LAUNCH(k_point_sweep<T>, bgrid_y((unsigned)((n + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK)),
                                       dim3(EMG_LINE_BLOCK), 0, stream, a);
                }
            } else {
                const i64 tmin = 7, tmax = (L.nC[0] - 1) + 2 * (L.nC[1] - 1) + 4 * (L.nC[2] - 1);
                const i64 n = (L.nC[1] - 1) * (L.nC[2] - 1);
                for (i64 th = tmin; th <= tmax; ++th) {
                    a.mode = 1; a.t = iback ? tmax - (th - tmin) : th;
                    MG_LAUNCH(k_point_sweep<T>, bgrid_y((unsigned)((n + EMG_LINE_BLOCK - 1) / EMG_LINE_BLOCK)),
                                       dim3(EMG_LINE_BLOCK), 0, stream, a);
                }
            }
        }
        if (hl) convert_field(L, L.eW[1], L.e, 1, true);
        check_launch();
    }

    // solver.smoothing, solver.py:738-799
    void smoothing(Level<T>& L, int nu, int lr_dir) {
        const int lr = current_lr_dir(lr_dir, L.nC);
        if (lr == 0) smooth_point(L, nu);
        int dirs[3], nd = 0;
        if (lr == 1 || lr == 5 || lr == 6 || lr == 7) dirs[nd++] = 0;
        if (lr == 2 || lr == 4 || lr == 6 || lr == 7) dirs[nd++] = 1;
        if (lr == 3 || lr == 4 || lr == 5 || lr == 7) dirs[nd++] = 2;
        // consecutive directions that share a working copy (y and z) convert once
        for (int k = 0; k < nd; ++k) {
            const bool in = (k == 0) || work_id(L, dirs[k - 1]) != work_id(L, dirs[k]);
            const bool out = (k == nd - 1) || work_id(L, dirs[k + 1]) != work_id(L, dirs[k]);
            smooth_line(L, dirs[k], nu, in, out);
        }
    }

    // ------------------------------------------------- residual / transfer
    // mode 1: L.r = s - A e ; mode 2: norm only -> norms[slot]
    void residual(Level<T>& L, int mode, int slot) {
        ResidualArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.nC[q] = L.nC[q]; a.eta[q] = L.eta[q]; a.h[q] = L.h[q]; a.ih[q] = L.ih[q]; }
        a.fl = L.fl; a.r = L.r; a.s = L.s; a.e = L.e; a.zeta = L.zeta; a.bt = batch(L);
        if (!dry && L.e_home == 1) {        // level 0, field and source in the x-split copies (home_on)
            if (!L.sW_valid[1]) { convert_field(L, L.sW[1], L.s, 1, true); L.sW_valid[1] = true; }
            a.e = L.eW[1]; a.s = L.sW[1]; a.xs = 1;
        }
        const i64 plane = (L.nC[0] + 1) * (L.nC[1] + 1);
        // levels with millions of cells: KZ node planes per thread (k_residual_zm, same results); below, a plane each
        const i64 cells = L.nC[0] * L.nC[1] * L.nC[2];
        const int kz = cells < res_zm_min ? 1 : res_kz ? res_kz : (cells * nsys >= ((i64)8 << 20) ? 8 : 4);
        const i64 nNz = L.nC[2] + 1;
        dim3 grid((unsigned)((plane + EMG_BLOCK - 1) / EMG_BLOCK), (unsigned)((nNz + kz - 1) / kz), (unsigned)nsys);
        a.nbx = grid.x;
        a.xcd = ((int)grid.x < res_xcd_min) ? 0 : res_xcd >= 0 ? res_xcd : (kz > 1 ? 2 : 1);
        if (a.xcd == 2) grid.x = 8 * ((grid.x + 7) / 8);       // strips: surplus workgroups exit
        const i64 np = (i64)a.nbx * nNz;                       // one partial per block and plane, whatever kz
        if (mode == 2 && np * nsys > n_partials) { partials = dalloc<double>(np * nsys); n_partials = np * nsys; }
        if (dry) return;
        {
            const char* tn = sizeof(T) == 16 ? "c128" : "f64";
            if (kz > 1) snprintf(res_name, sizeof res_name, "k_residual_zm<%s,%d,%d>", tn, mode, kz);
            else snprintf(res_name, sizeof res_name, "k_residual<%s,%d>", tn, mode);
        }
        if (mode == 2) {
            // a frozen system writes no partials: its norm slot keeps... nothing meaningful -- the host ignores
            // the norms of frozen systems (the partials are zeroed so that the sum stays finite)
            a.partials = partials;
            if (nsys > 1 && bmask) hipMemsetAsync(partials, 0, (size_t)(np * nsys) * sizeof(double), stream);
            if (!broken) residual_launch<T>(2, kz, grid, stream, a);
            MG_LAUNCH(k_sum_sqrt, dim3(nsys), dim3(EMG_BLOCK), 0, stream, (const double*)partials, np,
                               norm_out ? norm_out : norms, slot);
        } else {
            a.partials = nullptr;
            if (!broken) residual_launch<T>(1, kz, grid, stream, a);
        }
        check_launch();
    }

    void restrict_to(Level<T>& L, const Transfer& X, Level<T>& C) {   // solver.py:886-899
        const bool hl = home_lvl(C);
        if (hl) { ensure_work(C, 1); C.e_home = 1; }
        if (dry) return;
        RestrictArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.cnC[q] = C.nC[q]; a.fnC[q] = L.nC[q]; a.co[q] = X.co[q]; }
        a.cfl = C.fl; a.ffl = L.fl; a.cr = C.s; a.r = L.r; a.pec = 1;
        C.sT_valid = false; C.sW_valid[0] = C.sW_valid[1] = false;
        for (int ax = 0; ax < 3; ++ax) for (int q = 0; q < 3; ++q) a.w[ax][q] = X.w[ax][q];
        a.ce = C.e; a.bt = batch(L); a.cbst = C.nE;
        if (hl) { a.cr = C.sW[1]; a.ce = C.eW[1]; a.cxs = 1; C.sW_valid[1] = true; }
        i64 nmax = 0;       // one launch: blockIdx.y = component
        for (int c = 0; c < 3; ++c) {
            i64 n = 1;
            for (int q = 0; q < 3; ++q) n *= (q == c) ? C.nC[q] : C.nC[q] + 1;
            nmax = std::max(nmax, n);
        }
        MG_LAUNCH(k_restrict<T>, dim3((unsigned)((nmax + EMG_BLOCK - 1) / EMG_BLOCK), 3, (unsigned)nsys), dim3(EMG_BLOCK), 0, stream, a);
        check_launch();
    }

    void prolong_from(Level<T>& L, const Transfer& X, Level<T>& C) {  // solver.py:904-977
        if (dry) return;
        ProlongArgs<T> a;
        for (int q = 0; q < 3; ++q) { a.fnC[q] = L.nC[q]; a.cnC[q] = C.nC[q]; a.co[q] = X.co[q]; a.idx[q] = X.pidx[q]; a.wt[q] = X.pwt[q]; }
        a.ffl = L.fl; a.cfl = C.fl; a.e = L.e; a.ce = C.e; a.bt = batch(L); a.cbst = C.nE;
        if (L.e_home == 1) { a.e = L.eW[1]; a.fxs = 1; }
        if (C.e_home == 1) { a.ce = C.eW[1]; a.cxs = 1; }
        i64 nmax = 0;       // one launch: blockIdx.y = component
        for (int c = 0; c < 3; ++c) {
            i64 n = 1;
            for (int q = 0; q < 3; ++q) n *= (q == c) ? L.nC[q] : L.nC[q] + 1;
            nmax = std::max(nmax, n);
        }
        MG_LAUNCH(k_prolong<T>, dim3((unsigned)((nmax + EMG_BLOCK - 1) / EMG_BLOCK), 3, (unsigned)nsys), dim3(EMG_BLOCK), 0, stream, a);
        check_launch();
    }

    // ----------------------------------------------------------- recursion
    // One visit of solver.multigrid for level > 0 (solver.py:471-586).
    void mg_level(int g, int lr_dir, int level, int new_cycmax) {
        int cm;
        if (level == clevel[g]) cm = 1;
        else if (new_cycmax == 0 || cycle != 'F') cm = cycmax;
        else cm = new_cycmax;
        int cyc = 0, it = 0;
        while (it < cm) {
            iterate(g, lr_dir, level, cm - cyc, it, cm);
            ++it; ++cyc;
        }
    }

    // Body of the while loop (solver.py:524-577) for any level.
    void trace_point(Level<T>& L, int it, int level, int cm, int kind) {
        if (!trace || dry || nsys != 1) return;
        if ((int)trace_recs.size() >= TRACE_MAX) { ++trace_dropped; return; }
        if (!trace_norms) trace_norms = dalloc<double>(TRACE_MAX);
        if (!trace_norms) return;
        double* const keep = norm_out;
        norm_out = trace_norms;
        residual(L, 2, (int)trace_recs.size());
        norm_out = keep;
        trace_recs.push_back(TraceRec{it, level, cm, kind, {L.nC[0], L.nC[1], L.nC[2]}});
    }
    // it, cm: iteration count and cycmax of this level's loop (the trace reports them; level 0: it = -1, the host counts)
    void iterate(int g, int lr_dir, int level, int child_cycmax, int it = -1, int cm = 0) {
        Hierarchy<T>& H = hierarchy(g);
        Level<T>& L = *H.lv[level];
        if (level == clevel[g]) {
            smoothing(L, nu_coarse, lr_dir);
            trace_point(L, it, level, cm, 0);
        } else {
            if (nu_pre > 0) { smoothing(L, nu_pre, lr_dir); trace_point(L, it, level, cm, 1); }
            Level<T>& C = *H.lv[level + 1];
            residual(L, 1, 0);
            restrict_to(L, H.tr[level], C);
            mg_level(g, lr_dir, level + 1, child_cycmax);
            prolong_from(L, H.tr[level], C);
            if (nu_post > 0) { smoothing(L, nu_post, lr_dir); trace_point(L, it, level, cm, 2); }
        }
    }

    // One level-0 iteration + end-of-cycle residual norm into norms[slot].
    // cycmax of LEVEL 0.  solver.multigrid evaluates it ONCE, on entry, with the sc_dir of that moment
    // (solver.py:480-485), and keeps it while sc_dir rotates from cycle to cycle: a solve that starts with a direction
    // whose clevel is 0 (level 0 is its coarsest grid, e.g. 8 x 3 x 3 with sc_dir 1) hands new_cycmax = 1 to the
    // children of ALL its later cycles -- its F-cycles then visit the coarse levels once, like V-cycles.
    // emg3d_mg_begin(sc_dir) records that moment; 0: not recorded, every cycle uses its own sc_dir.
    int entry_cm = 0;
    void cycle0_eager(int g, int lr_dir, int slot) {
        int cm = entry_cm ? entry_cm : ((0 == clevel[g]) ? 1 : cycmax);   // level 0: new_cycmax == 0
        iterate(g, lr_dir, 0, cm, -1, cm);        // cyc == 0 on level 0 (solver.py:585-586)
        residual(*lv0, 2, slot);
    }

    // A cycle is ~10^3 dependent kernel launches, many of them a few
    // microseconds long: issued one by one the host cannot keep the queue full.
    // The launch sequence of a (sc_dir, lr_dir) pair is fixed once its hierarchy
    // and factor caches exist, so it is captured into a hipGraph on its second
    // use and replayed afterwards (the first use runs eagerly and allocates).
    // The level-0 source changes only through set_sfield; its working copies are
    // refreshed here, OUTSIDE the captured graphs (a graph captured while a copy
    // was still valid would otherwise replay without the conversion).
    void refresh_level0_source() {
        Level<T>& L = *lv0;
        if (L.sT && !L.sT_valid) { convert_field(L, L.sT, L.s, -1, true); L.sT_valid = true; }
        for (int w = 0; w < 2; ++w)
            if (L.sW[w] && !L.sW_valid[w]) { convert_field(L, L.sW[w], L.s, w, true); L.sW_valid[w] = true; }
        for (int d = 0; d < 3; ++d)
            if (L.sflag[d]) ensure_sflags(L, d);
    }

    // Everything a cycle with (sc_dir g, lr_dir) needs that is loop invariant: grid hierarchy, transfer
    // weights, coarse models, factor caches, work buffers and the captured launch sequence.  Launches no
    // cycle; the fields are untouched.
    void prepare(int g, int lr_dir) {
        if (!use_graph) { dry = true; cycle0_eager(g, lr_dir, 0); dry = false; place_level0(lr_dir); return; }
        cycle0(g, lr_dir, -1);
    }

    // slot < 0: prepare only (see above)
    void cycle0(int g, int lr_dir, int slot) {
        if (!use_graph || trace) {      // (tracing: eager launches)
            if (slot >= 0) {
                if (placement_pending(lr_dir)) prepare(g, lr_dir);      // (eager cycles place their working copies too)
                cycle0_eager(g, lr_dir, slot);
            }
            return;
        }
        const int key = (g * 8 + lr_dir) * 4 + entry_cm;       // the captured launch sequence depends on level 0's cycmax
        auto it = graphs.find(key);
        if (it != graphs.end() && slot < 0) return;
        // a cycle that line-smooths level 0 starts (and ends) with the field in the x-split copy
        const int home_in = (home_on(*lv0) && current_lr_dir(lr_dir, lv0->nC) != 0) ? 1 : 0;
        if (it == graphs.end()) {
            // dry run: build hierarchy, factor caches and work buffers (the only
            // steps that allocate), then capture the launch sequence
            const bool tlog = getenv("EMG3D_LOG_SETUP") != nullptr;
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            auto t0 = now();
            const size_t nalloc0 = allocs.size();
            dry = true;
            cycle0_eager(g, lr_dir, 0);
            dry = false;
            place_level0(lr_dir);           // (large level 0: choose the blocks its sweeps write to, before any graph holds them)
            // where the captured sequence finds the field.  Launch path: move it there now.  Prepare-only path (it may run
            // on the side stream beside a cycle that is using the field): nothing is moved, the branch decisions of the
            // capture are taken as if, and the real state comes back afterwards -- the launch converts when it is due.
            const int home_keep = (slot >= 0) ? home_in : lv0->e_home;
            if (slot >= 0) { if (home_in) e_to_w1(*lv0); else e_to_ref(*lv0); }
            else if (home_in) ensure_work(*lv0, 1);
            lv0->e_home = home_in;
            refresh_level0_source();
            if (tlog) hipStreamSynchronize(stream);
            auto t1 = now();
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            hipError_t st = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
            if (st == hipSuccess) {
                cycle0_eager(g, lr_dir, 0);
                st = hipStreamEndCapture(stream, &graph);
            }
            const int home_out = lv0->e_home;
            lv0->e_home = home_keep;        // nothing has run yet
            auto t2 = now();
            if (st == hipSuccess) st = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) hipGraphDestroy(graph);
            if (tlog) fprintf(stderr, "[setup] key %d: hierarchy+factors %.2f ms (%zu allocations), capture %.2f ms, instantiate %.2f ms\n",
                              key, ms(t0, t1), allocs.size() - nalloc0, ms(t1, t2), ms(t2, now()));
            if (st != hipSuccess || err != 0) {
                // capture not possible: fall back to eager launches for good
                (void)hipGetLastError();
                fprintf(stderr, "[emg3d_hip] hipGraph capture failed (%s); using eager launches\n", hipGetErrorString(st));
                use_graph = false;
                err = 0;
                if (slot >= 0) cycle0_eager(g, lr_dir, slot);
                return;
            }
            (void)hipGraphUpload(exec, stream);     // move the first-launch upload out of the first cycle
            graphs[key] = exec;
            graph_home[key] = std::make_pair(home_in, home_out);
            it = graphs.find(key);
            if (slot < 0) return;
        }
        const std::pair<int, int> hm = graph_home[key];
        if (hm.first) e_to_w1(*lv0); else e_to_ref(*lv0);
        refresh_level0_source();
        hipError_t st = hipGraphLaunch(it->second, stream);
        if (st != hipSuccess && err == 0) err = (int)st;
        lv0->e_home = hm.second;
        if (slot != 0) hipMemcpyAsync(norms + (i64)slot * nsys, norms, (size_t)nsys * sizeof(double), hipMemcpyDeviceToDevice, stream);
    }

    void forget_factors() {
        // (the launch descriptors carry factor offsets of the layout they were generated for: they go with the factor)
        auto clear = [](Level<T>& L) {
            for (int d = 0; d < 3; ++d) {
                L.fac[d] = nullptr; L.fac_kind[d] = 0;
                for (int c = 0; c < 4; ++c) { L.qd[d][c] = nullptr; L.qdn[d][c] = 0; }
            }
        };
        if (lv0) clear(*lv0);
        for (auto& kv : hier) for (auto& l : kv.second.lv) if (l) clear(*l);
    }

    void drop_graphs() {
        for (auto& kv : graphs) hipGraphExecDestroy(kv.second);
        graphs.clear();
        graph_home.clear();
        graph_seen.clear();
    }
};
