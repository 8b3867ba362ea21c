/*
 * emg3d_hip.h -- C ABI of the MI355X (gfx950) multigrid hot path for emg3d.
 *
 * This shared library (libemg3d_hip.so) stands where the numba module
 * `emg3d/core.py` and the cycle sub-routines of `emg3d/solver.py` stand in the
 * reference (emg3d v0.17.0).  Plain pointers and sizes only; no torch types.
 *
 * Conventions (identical to the reference):
 *   dtype   0 = float64 (Laplace domain), 1 = complex128 (frequency domain;
 *           interleaved re,im doubles)                     fields.py:417-420
 *   field   ONE 1-D buffer [fx | fy | fz]; fx F-ordered (nCx,nNy,nNz),
 *           fy (nNx,nCy,nNz), fz (nNx,nNy,nCz)              fields.py:253-281
 *   eta_*   F-ordered (nCx,nCy,nCz) of `dtype`; eta_y/eta_z may alias eta_x
 *   zeta,h  float64                                         models.py:631-658
 *   order   0 = lexicographic (the reference's update order, executed as
 *               hyperplane wavefronts; same result as the sequential sweep)
 *           1 = multi-colour (4 colours for line smoothers, 8 for the point
 *               smoother): the throughput mode
 * Every function returns 0 on success, a HIP error code (>0) on a device
 * error, or a negative value for invalid arguments; numerical blow-ups surface
 * as NaN/Inf in the data exactly as in the reference (solver.py:1715).
 *
 * Tier 1: stateless host-pointer kernels, one per `emg3d.core` function
 * (H2D, kernel, D2H inside the call) -- used by the drop-in `core` module and
 * the parity tests.  Tier 2: a handle that keeps grids, model, fields and the
 * cached line factorisations device-resident across a whole solve.
 */
#ifndef EMG3D_HIP_H
#define EMG3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / device ------------------------------------------------- */
/* ABI version: bumped whenever an entry point is added or a signature changes; emg3d_hip_version() returns the value the
 * library was built with, and the Python binding (emg3d_amd/_lib.py: ABI_VERSION) refuses a library of another version. */
#define EMG3D_HIP_ABI_VERSION 103
int emg3d_hip_version(void);
int emg3d_hip_device_count(int* count);
int emg3d_hip_set_device(int device);
/* name must hold >= 256 bytes */
int emg3d_hip_device_info(int device, char* name, int64_t* total_mem, int* cu_count);
/* bytes the driver reports free / in total on `device` right now (hipMemGetInfo); blocks parked in this process's own
 * pool (emg3d_hip_cached_bytes) count as used there although the next handle may take them                          */
int emg3d_hip_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);

/* Device blocks of destroyed handles are kept for the next handle of the process (exact-size reuse, bounded by
 * EMG3D_POOL_GB, default 96; 0 disables): release them to the driver / ask how much is parked.            */
int64_t emg3d_hip_release_cached(void);
int64_t emg3d_hip_cached_bytes(void);
/* ... of which DEVICE memory parked for `device` (the pool also holds blocks of other devices of the process and pinned host
 * staging buffers, neither of which an allocation on `device` can take)                                              */
int64_t emg3d_hip_cached_bytes_on(int device);

/* ---- Tier 1: `emg3d.core` equivalents on host pointers ----------------- */

/* core.amat_x(rx,ry,rz, ex,ey,ez, eta_x,eta_y,eta_z, zeta, hx,hy,hz)
 * reference emg3d/core.py:29-177.  r -= A e, in place in r.               */
int emg3d_amat_x(int dtype, int64_t nx, int64_t ny, int64_t nz, void* r, const void* e,
                 const void* eta_x, const void* eta_y, const void* eta_z, const double* zeta,
                 const double* hx, const double* hy, const double* hz);

/* fields.get_h_field(grid, model, field), reference emg3d/fields.py:819-911: H = -curl E / (s mu_0) on the
 * faces, [hx|hy|hz] F-ordered (nNx,nCy,nCz), (nCx,nNy,nCz), (nCx,nCy,nNz).  zeta = V/mu_r when the model has
 * mu_r (fields.py:878-906), NULL otherwise.  (smu0_re, smu0_im) = field.smu0; real dtype: smu0_im = 0.   */
int emg3d_get_h_field(int dtype, int64_t nx, int64_t ny, int64_t nz, void* hfield, const void* efield,
                      const double* zeta, const double* hx, const double* hy, const double* hz,
                      double smu0_re, double smu0_im);

/* core.gauss_seidel (dir=0, core.py:181-474), core.gauss_seidel_x/_y/_z
 * (dir=1/2/3, core.py:477-753, 756-1037, 1040-1316): nu sweeps, in place. */
int emg3d_gauss_seidel(int dtype, int dir, int64_t nx, int64_t ny, int64_t nz, void* e,
                       const void* s, const void* eta_x, const void* eta_y, const void* eta_z,
                       const double* zeta, const double* hx, const double* hy, const double* hz,
                       int nu, int order);

/* core.restrict(crx,cry,crz, rx,ry,rz, wx,wy,wz, sc_dir), core.py:1586-1967.
 * w = [wxl,wx0,wxr, wyl,wy0,wyr, wzl,wz0,wzr] (host pointers).             */
int emg3d_restrict(int dtype, int64_t nx, int64_t ny, int64_t nz, int64_t cnx, int64_t cny,
                   int64_t cnz, void* cr, const void* r, const double* const* w, int sc_dir);

/* core.restrict_weights(vectorN, vectorCC, h, cvectorN, cvectorCC, ch),
 * core.py:1970-2041.  nh = len(h), n = len(cvectorN); O(n) host work.      */
int emg3d_restrict_weights(const double* vectorN, const double* vectorCC, const double* h,
                           int64_t nh, const double* cvectorN, const double* cvectorCC,
                           const double* ch, int64_t n, double* wl, double* w0, double* wr);

/* core.solve(amat, bvec), core.py:1447-1582: banded LDL^T, n unknowns.     */
int emg3d_solve(int dtype, void* amat, void* bvec, int64_t n);

/* core.blocks_to_amat(amat,bvec,middle,left,rhs,im,nC), core.py:1319-1444.
 * n = number of unknowns (len(bvec)); left is float64 as in the reference. */
int emg3d_blocks_to_amat(int dtype, void* amat, void* bvec, int64_t n, const void* middle,
                         const double* left, const void* rhs, int64_t im, int64_t nC);

/* solver.prolongation(grid, efield, cgrid, cefield, sc_dir),
 * solver.py:904-977 (+ RegularGridProlongator 1368-1463): e += P ce, then
 * ensure_pec.  Fine grid (nx,ny,nz; hx,hy,hz); sc_dir in 0..6.             */
int emg3d_prolongation(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                       const double* hy, const double* hz, const double* origin, void* e,
                       const void* ce, int sc_dir);

/* solver._restrict_model_parameters(param, sc_dir), solver.py:1747-1784.
 * is_complex selects the scalar type of param (eta: dtype, zeta: 0).       */
int emg3d_restrict_model(int is_complex, int64_t nx, int64_t ny, int64_t nz, void* cparam,
                         const void* param, int sc_dir);

/* Which line-sweep kernel the library selects for the colour launches (order 1; order 0: the hyperplane launches) of a level of
 * nx x ny x nz cells along dir (1, 2, 3 = x, y, z) on a device of cu_count compute units (<= 0: the current device) with nsys
 * batched systems -- the launch selection of the handle (reference: the one loop of core.gauss_seidel_x/_y/_z, emg3d/core.py:477-1316,
 * has no such choice) evaluated on the shape alone: no device memory, no launch, callable without a GPU when cu_count > 0.
 * name (>= 64 bytes): the instantiation as emg3d_mg_last_sweep_kernel reports it; info[6]: lines of the largest colour, lines per
 * wave (thm: per pair of waves, tha / qpl: per workgroup), rounds of waves of that colour's launch, factor layout (0 one-sided 15
 * numbers per block, 3 mirrored two-sided, 4 compact 11 numbers), parity-split working copies (0 / 1), 64-bit field offsets (0 / 1). */
int emg3d_sweep_plan(int dtype, int64_t nx, int64_t ny, int64_t nz, int dir, int order, int nsys, int cu_count, char* name,
                     int64_t* info);

/* ---- Tier 2: device-resident multigrid handle --------------------------- */
typedef struct emg3d_mg emg3d_mg_t;

/* Builds the level-0 state on `device`: uploads h, eta, zeta; origin[3]
 * (or NULL = 0) is the grid origin (node coordinates enter the transfer
 * weights exactly as in the reference, solver.py:859-864).
 * (VolumeModel outputs, models.py:554-658, are inputs here.)               */
int emg3d_mg_create(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz,
                    const double* hx, const double* hy, const double* hz, const double* origin,
                    const void* eta_x, const void* eta_y, const void* eta_z, const double* zeta,
                    int device);
/* Same handle from the FREQUENCY-INDEPENDENT model: sv_* = conductivity * cell volume (real,
 * F-ordered (nCx,nCy,nCz); sv_y / sv_z may alias sv_x or be NULL) and the scalar
 * smu0 = s*mu_0 (= -2 pi i f mu_0, or f mu_0 in the Laplace domain; imaginary part ignored for
 * dtype 0): eta = smu0 * sv is formed on the device (models.py:631-658 without epsilon_r).  The
 * ranks of a frequency shard (simulations.py:840-867) share sv and differ in one scalar.      */
int emg3d_mg_create_sv(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                       const double* hy, const double* hz, const double* origin, const double* sv_x,
                       const double* sv_y, const double* sv_z, const double* zeta, double smu0_re,
                       double smu0_im, int device);

/* The same with the conductivities and the cell volumes as separate arrays: eta = (s mu_0 V) sigma is formed on the device
 * exactly as VolumeModel rounds it (reference models.py:631-658, `(smu0 * vol) * sigma`), so that a handle -- also one
 * re-targeted with emg3d_mg_set_smu0 -- holds bit for bit the eta of the reference at every frequency.  s mu_0 must be
 * purely imaginary (dtype 1) or real (dtype 0): -2 otherwise.  resistivity != 0: the sigma arrays hold resistivities
 * (Model's 'Resistivity' mapping) and the device takes the reciprocal (an IEEE division: Model.conductivity's bits).  */
int emg3d_mg_create_vs(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                       const double* hy, const double* hz, const double* origin, const double* sigma_x,
                       const double* sigma_y, const double* sigma_z, const double* vol, const double* zeta,
                       double smu0_re, double smu0_im, int resistivity, int device);
/* ... and with displacement currents (Model.epsilon_r): eta = s mu_0 V (sigma - s eps_0 eps_r), reference
 * models.py:639-647, formed on the device as NumPy rounds it.  seps0 = s eps_0 (dtype 0) resp. Im(s) eps_0 (dtype 1: s is
 * purely imaginary, and so is s mu_0 = i smu0_im).  Another frequency: emg3d_mg_set_smu0_eps.                        */
int emg3d_mg_create_vse(emg3d_mg_t** out, int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx,
                        const double* hy, const double* hz, const double* origin, const double* sigma_x,
                        const double* sigma_y, const double* sigma_z, const double* vol, const double* zeta,
                        const double* epsilon_r, double smu0_re, double smu0_im, double seps0, int resistivity, int device);
void emg3d_mg_destroy(emg3d_mg_t* mg);

/* Cycle parameters = the MGParameters fields used inside solver.multigrid
 * (solver.py:1043-1113): cycle 'V'/'W'/'F' as char code, nu_*, clevel[4] =
 * MGParameters.clevel after max_level (solver.py:1142-1173), order as above. */
int emg3d_mg_set_params(emg3d_mg_t* mg, int cycle, int nu_init, int nu_pre, int nu_coarse,
                        int nu_post, const int* clevel, int order);

int emg3d_mg_set_sfield(emg3d_mg_t* mg, const void* sfield_host);
/* Source from the real, frequency-independent source vector: s = smu0 * vector (fields.py:624,
 * `SourceField.vector`); uploads 8 instead of 16 bytes per edge.  Overwrites the residual buffer. */
int emg3d_mg_set_sfield_vector(emg3d_mg_t* mg, const double* vector, double smu0_re, double smu0_im);
/* Source of ONE finite electric dipole src6 = (x0, x1, y0, y1, z0, z1) built in HBM (fields.get_source_field for
 * a finite dipole, reference fields.py:586-629; the edge distribution _finite_source_xyz, fields.py:914-1010, runs
 * on the device): s (+)= scale_c * weights_c per component, scale6 = (re, im) x 3 = moment_c * s mu_0 (fields.py:624;
 * imaginary parts ignored by float64 handles); coordinates and nodes are rounded to `decimals` as in the reference;
 * accumulate != 0 adds to the present source (arbitrarily shaped dipoles, fields.py:575-580).  sums3 (may be NULL)
 * receives the three weight sums (1 for a source inside the grid; |sum - 1| > 1e-6 triggers the reference's
 * normalisation).  Returns -4 when the source lies outside the grid (the reference raises ValueError).
 * A point dipole [x, y, z, azimuth, dip] is turned into a finite one by the caller (fields.py:1037-1040).     */
int emg3d_mg_set_sfield_dipole(emg3d_mg_t* mg, const double* src6, const double* scale6, int decimals,
                               int accumulate, double* sums3);
/* The same as a stateless call that returns the source field on the host ([fx|fy|fz], nE entries of dtype). */
int emg3d_source_field(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx, const double* hy,
                       const double* hz, const double* origin, const double* src6, const double* scale6,
                       int decimals, void* sfield, double* sums3);
int emg3d_mg_set_efield(emg3d_mg_t* mg, const void* efield_host); /* NULL -> zeros */
int emg3d_mg_get_efield(emg3d_mg_t* mg, void* efield_host);
int emg3d_mg_get_residual(emg3d_mg_t* mg, void* rfield_host);     /* r = s - A e */
/* fields.get_h_field (fields.py:819-911) of the device-resident level-0 electric field; use_zeta != 0 when
 * the model has mu_r.  48 instead of 96+ bytes per cell over PCIe.  Overwrites the residual buffer.      */
int emg3d_mg_get_hfield(emg3d_mg_t* mg, int use_zeta, double smu0_re, double smu0_im, void* hfield_host);

/* ---- receivers ----------------------------------------------------------------------
 * maps.interp3d(points, values, new_points, method, fill_value, mode='constant', cval), reference
 * emg3d/maps.py:179-276, on the device: values F-ordered (nx,ny,nz) of dtype on the regular grid px,py,pz;
 * xi = [x[n] | y[n] | z[n]]; method 0 = 'linear' (RegularGridInterpolator, bounds_error=False; has_fill = 0 means
 * fill_value=None: extrapolate), 1 = 'cubic' (the SciPy arithmetic the reference calls: not-a-knot index spline +
 * cubic B-spline prefilter + 4x4x4 evaluation; points outside get cval, complex: cval + cval j); fewer than 4 points
 * along an axis force 'linear' (maps.py:238-240); 2 / 3 / 4 = 'cubic' with xi already in INDEX coordinates of `values`
 * (px, py, pz unused): how the binding serves map_coordinates' boundary modes 'mirror' (2: stencil indices mirrored;
 * also 'wrap', whose coordinates the caller wraps with period n - 1 first: SciPy's legacy rule), 'nearest' (3: values
 * edge-padded by 12 samples by the caller; "reflect" prefilter initialisation, stencil indices clamped to the array) and
 * 'reflect' (4: "reflect" prefilter initialisation, stencil indices reflected); no point is outside in any of them -- the
 * O(n) index spline on the host, prefilter and evaluation over the whole array here.  The stateless entry points run on
 * the calling thread's current device.                                                                                    */
int emg3d_interp3d(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* px, const double* py,
                   const double* pz, const void* values, int64_t n, const double* xi, int method, int has_fill,
                   double fill_value, double cval, void* out);
/* fields.get_receiver_response(grid, field, rec), reference emg3d/fields.py:733-817: the field at n point
 * receivers, resp[r] = sum_c factors[c][r] * interp3d(points_c[1:-1], field_c[1:-1,1:-1,1:-1], (x,y,z)[r], 'cubic',
 * 0.0, 'constant', nan).  xyz = [x[n] | y[n] | z[n]]; factors = [fx[n] | fy[n] | fz[n]] = fields._rotation(azimuth,
 * dip) (fields.py:1013-1034, evaluated by the caller); a component whose factors are all <= 1e-10 is skipped
 * (fields.py:810).  field: [fx|fy|fz] of an electric (is_electric != 0) or magnetic field on the host.        */
int emg3d_get_receiver_response(int dtype, int64_t nx, int64_t ny, int64_t nz, const double* hx, const double* hy,
                                const double* hz, const double* origin, const void* field, int is_electric,
                                int64_t n, const double* xyz, const double* factors, void* resp);
/* Same on the device-resident level-0 electric field of a handle (magnetic != 0: on H = get_h_field(E),
 * fields.py:819-911, formed on the device; use_zeta / smu0 as in emg3d_mg_get_hfield): 16 bytes per receiver
 * cross PCIe instead of the field.  Overwrites the residual buffer when magnetic != 0.                      */
int emg3d_mg_get_receiver_response(emg3d_mg_t* mg, int magnetic, int use_zeta, double smu0_re, double smu0_im,
                                   int64_t n, const double* xyz, const double* factors, void* resp);

/* ---- adjoint-state gradient ------------------------------------------------------------
 * maps.edges2cellaverages(ex, ey, ez, vol, out_x, out_y, out_z), reference emg3d/maps.py:578-630: edge values to
 * volume-weighted cell averages, ADDED into out_x/y/z (F-ordered (nx,ny,nz) of dtype, host); field = [fx|fy|fz].  */
int emg3d_edges2cellaverages(int dtype, int64_t nx, int64_t ny, int64_t nz, const void* field, const double* vol,
                             void* out_x, void* out_y, void* out_z);
/* optimize.gradient for ONE (source, frequency) pair on its computational grid, reference
 * emg3d/optimize.py:176-199: grad = sum_c edges2cellaverages_c(-Re(bfield * efield * smu0)) with the cell volumes
 * of the handle's grid.  bfield = the handle's level-0 field (the back-propagated solution, simulations.py:1131-1143),
 * efield = workspace vector `efield_vec` (the forward solution, saved with emg3d_mg_vec_copy(id, -2)).  grad: nC
 * doubles, F-ordered.  (The reference then maps -grad to the model grid, maps.grid2grid: gridding, out of scope.)
 * Overwrites the residual buffer.                                                                                */
int emg3d_mg_gradient(emg3d_mg_t* mg, int efield_vec, double smu0_re, double smu0_im, double* grad);

/* ---- batched systems: several sources through the same launches ------------------------------------------
 * The reference solves one (source, frequency) system per solver.solve call; a survey has many sources per
 * frequency on ONE model (simulations.py:916-1015 loops over them, one job per source-frequency pair).  Systems
 * that share grid, model and frequency share the operator and the cached line factorisations, so a handle can carry
 * n of them through every launch of the cycle (arrays [system][nE]; kernels index the system by a grid dimension):
 * the ~1100 latency-bound coarse-level launches of a cycle do n times the work and the level-0 sweeps fetch the
 * factor once for n right-hand sides.  Each system goes through exactly the arithmetic of a solve of its own --
 * results are bit-identical to n separate handles.
 *   emg3d_mg_set_batch(mg, n)   1 <= n <= 64, before the first cycle / prepare (returns -6 afterwards); zeroes
 *                               the fields.
 *   emg3d_mg_select(mg, b)      the system that the single-field entry points address from now on: set/get
 *                               sfield/efield, set_sfield_dipole/vector, get_hfield, get_receiver_response,
 *                               get_residual, sfield_norm, gradient, *_devptr.
 *   emg3d_mg_set_mask(mg, act)  act[n]: 0 freezes a system (converged: its cycles are skipped, its field stays
 *                               untouched); the norms reported for a frozen system are 0.
 * emg3d_mg_cycle / emg3d_mg_residual_norm then write n norms, emg3d_mg_cycles ncycles x n ([cycle][system]).
 * The Krylov workspace (emg3d_mg_vec_*) addresses the selected system only.  Environment EMG3D_BATCH_TUNE=1 (read at
 * emg3d_mg_create): coarse-level kernel choice by lines x systems -- faster (6-11 %), results then equal stand-alone
 * solves to rounding instead of bit for bit.                                                                      */
int emg3d_mg_set_batch(emg3d_mg_t* mg, int n);
int emg3d_mg_get_batch(emg3d_mg_t* mg);
int emg3d_mg_select(emg3d_mg_t* mg, int b);
int emg3d_mg_set_mask(emg3d_mg_t* mg, const int* active);

/* solver.residual(..., norm=True), solver.py:980-1039 on the level-0 state. */
int emg3d_mg_residual_norm(emg3d_mg_t* mg, double* l2);
/* ||sfield||_2 (solver.py:305). */
int emg3d_mg_sfield_norm(emg3d_mg_t* mg, double* l2);

/* solver.smoothing on the level-0 state (solver.py:738-799). */
int emg3d_mg_smooth(emg3d_mg_t* mg, int nu, int lr_dir);

/* Another frequency on the same handle (handles made by emg3d_mg_create_sv / emg3d_mg_create_vs): eta is re-formed from the
 * sigma*V (or sigma and V) kept in HBM, the coarse models of every hierarchy built so far, the transposed model copies and every cached
 * line factorisation are recomputed by the kernels a fresh handle would run -- bit for bit a fresh handle's results --,
 * while grids, transfer weights, work buffers and captured launch graphs stay (the per-frequency jobs of
 * Simulation.compute, emg3d/simulations.py:840-867, share everything but this scalar; models.py:631-658).
 * -7: the handle was created from eta arrays; -2: complex s mu_0 for a float64 handle.                            */
int emg3d_mg_set_smu0(emg3d_mg_t* mg, double smu0_re, double smu0_im);
/* The same for handles made by emg3d_mg_create_vse (-7 for any other; emg3d_mg_set_smu0 answers -7 for these). */
int emg3d_mg_set_smu0_eps(emg3d_mg_t* mg, double smu0_re, double smu0_im, double seps0);

/* Entry of solver.multigrid (solver.py:471-492): the reference fixes the cycmax of level 0 when the function is
 * entered, from var.clevel[var.sc_dir] of THAT moment, and keeps it for all cycles of the call although sc_dir rotates
 * (semicoarsening=True or several digits).  It matters when the first direction has clevel 0 (level 0 is its coarsest
 * grid: small odd transverse sizes) and a later one has not: the children of every later F-cycle then get
 * new_cycmax = 1.  Call once per multigrid() call, after emg3d_mg_set_params, with the current sc_dir; without it every
 * cycle derives level 0's cycmax from its own sc_dir.                                                             */
int emg3d_mg_begin(emg3d_mg_t* mg, int sc_dir);

/* ONE level-0 iteration of solver.multigrid (solver.py:518-591): pre-smooth,
 * residual, restriction, recursive coarse-grid correction (V/W/F), prolongation,
 * post-smooth, and the end-of-cycle residual norm.  sc_dir/lr_dir are the
 * current var.sc_dir / var.lr_dir.  No host<->device traffic except *l2.    */
int emg3d_mg_cycle(emg3d_mg_t* mg, int sc_dir, int lr_dir, double* l2);
/* The same cycle; while the device runs it the host does emg3d_mg_prepare(next_sc_dir, next_lr_dir) -- the pair the
 * solver's rotation (solver.py:597-600) uses in the NEXT cycle -- so that the set-up of the second and third pair of an
 * sc+lr run (5 ms each at 128^3) is hidden behind the first cycles.  next_* < 0: nothing to prepare.  Same results. */
int emg3d_mg_cycle_next(emg3d_mg_t* mg, int sc_dir, int lr_dir, int next_sc_dir, int next_lr_dir, double* l2);
/* verb = 5 of the reference (solver.py:502-578: the residual norm after every smoothing call of every level, printed as
 * "it level cycmax [nx, ny, nz]: norm  pre-smoothing" ...): with the trace on, emg3d_mg_cycle* run their launches eagerly
 * and follow every smoothing call with a norm-only residual; emg3d_mg_get_trace returns the records collected since the
 * last call -- recs[i] = {it, level, cycmax, kind (0 coarsest level, 1 pre-, 2 post-smoothing), nx, ny, nz}, it = -1 on
 * level 0 (the caller counts level-0 iterations) -- and their norms, in the order the reference would print them.
 * One system per handle.                                                                                            */
int emg3d_mg_set_trace(emg3d_mg_t* mg, int on);
int emg3d_mg_get_trace(emg3d_mg_t* mg, int max_recs, int64_t* recs, double* norms, int* count);
/* Loop-invariant set-up of the cycles with this (sc_dir, lr_dir): grid hierarchy, restriction /
 * prolongation weights, coarse models (solver.py:802-901, done once instead of per cycle), the
 * cached line factorisations and the captured launch sequence.  Optional -- the first cycle does
 * the same on demand -- and idempotent; runs no cycle and leaves the fields untouched.          */
int emg3d_mg_prepare(emg3d_mg_t* mg, int sc_dir, int lr_dir);

/* Same, `ncycles` times back to back with FIXED sc_dir/lr_dir rotation
 * sc_cycle/lr_cycle (arrays of length n_sc/n_lr, start positions given);
 * l2 receives ncycles norms.  Used by bench.py (no per-cycle host sync).    */
int emg3d_mg_cycles(emg3d_mg_t* mg, int ncycles, const int* sc_cycle, int n_sc, const int* lr_cycle,
                    int n_lr, double* l2);

/* Device pointers / stream for zero-copy interop (torch, RCCL).
 * emg3d_mg_efield_devptr: the level-0 field in the reference layout [fx|fy|fz].  Large levels keep the field in a
 * parity-split working copy between cycles; the call then enqueues the conversion on the handle's stream and returns
 * the reference-layout buffer.  The pointer is therefore a SNAPSHOT: valid (and ordered behind the handle's stream)
 * until the next cycle / smoothing / solve call on the handle; writes through it are not seen by later cycles
 * (use emg3d_mg_set_efield).  Fetch it again after every such call.  NULL on error.                               */
void* emg3d_mg_efield_devptr(emg3d_mg_t* mg);
void* emg3d_mg_sfield_devptr(emg3d_mg_t* mg);
void* emg3d_mg_stream(emg3d_mg_t* mg);
int64_t emg3d_mg_nE(emg3d_mg_t* mg);
int emg3d_mg_sync(emg3d_mg_t* mg);
/* Bytes of device memory currently held by the handle. */
int64_t emg3d_mg_device_bytes(emg3d_mg_t* mg);

/* Timing of the dominant kernel (line-smoother substitution sweep) with HIP
 * events on the handle's stream: runs `reps` sweeps (nu=1) in direction
 * dir (1,2,3) on the level-0 state; returns average ms per sweep.          */
int emg3d_mg_time_sweep(emg3d_mg_t* mg, int dir, int reps, float* ms_per_sweep);
/* Placement of level 0's working copies (levels whose working copy is >= 256 MiB, multi-colour order; EMG3D_PLACE_TRIES=<n>,
 * default 12, 0 = off): the duration of a level-0 colour launch at 256^3 depends by up to 10 % on which piece of physical
 * memory the block it WRITES got, so before the first launch sequence on working copy w (0: x-lines, 1: y-/z-lines) is
 * captured the handle times up to n candidate blocks with one sweep each, until it holds one of the fast class, and keeps the
 * fastest; when the handle goes the block is parked under its role and the next handle of the process takes it without a
 * search.  The record: *tries candidates timed (0: nothing was placed, or *kept = -1: a block of an earlier handle's search
 * was taken), *kept the index of the one that stayed (0 = the block the handle had), ms[k] = ms per sweep (4 launches) of
 * candidate k; ms must hold 16 floats.  Results do not depend on it.                                                     */
int emg3d_mg_placement(emg3d_mg_t* mg, int w, int* tries, int* kept, float* ms);
/* Name of the kernel instantiation that the handle's most recent line-sweep launch selected, e.g.
 * "k_line_sweep_th<c128,3,8>" (what `rocprofv3 --kernel-trace` shows); name must hold >= 64 bytes.      */
int emg3d_mg_last_sweep_kernel(emg3d_mg_t* mg, char* name);
/* Same for the residual kernel (amat_x).                                    */
int emg3d_mg_time_residual(emg3d_mg_t* mg, int reps, float* ms_per_call);
/* Name of the kernel instantiation of the handle's most recent residual launch, e.g. "k_residual_zm<c128,1,4>"
 * (node planes per thread and block map are chosen by level size; all give identical results); >= 64 bytes. */
int emg3d_mg_last_residual_kernel(emg3d_mg_t* mg, char* name);

/* Device-side Krylov building blocks for the BiCGSTAB path (solver.py:610-734):
 * y = A x (core.amat_x on a zero field, negated: solver.py:646-660) and
 * x = M b (one preconditioner application = `ncycles` MG cycles on a zero
 * field, solver.py:667-677) on host vectors.                                */
int emg3d_mg_amatvec(emg3d_mg_t* mg, const void* x_host, void* y_host);

/* Device-resident vector workspace for the Krylov iteration that the reference
 * delegates to scipy.sparse.linalg.bicgstab / cgs / gcrotmk (call site solver.py:717-719;
 * SciPy is pinned only as scipy>=1.4, setup.py:39): the vectors r, r~, p, v, s, t, p^, s^, x
 * (gcrotmk: the Krylov, preconditioned and recycled outer vectors) stay in HBM, only scalars
 * cross PCIe.  Vector ids: 0..n-1 = workspace vectors (nE entries of the handle's dtype,
 * zero-initialised by emg3d_mg_vec_alloc, which may be called again to grow the workspace; n <= 256),
 * -1 = the level-0 source s, -2 = the level-0 field e (so that a preconditioner
 * application is vec_copy(-1, b); set_efield(NULL); cycles; vec_copy(x, -2)).
 *   vec_axpy  : y += alpha x          vec_scale : y *= alpha
 *   vec_dot   : out2 = (Re, Im) of sum conj(a_i) b_i  (numpy.vdot; plain dot for float64),
 *               deterministic reduction; synchronises the stream
 *   vec_amatvec: dst = A src          (core.amat_x on a zero field, negated: solver.py:646-660)
 * alpha_im is ignored for float64 handles.                                   */
int emg3d_mg_vec_alloc(emg3d_mg_t* mg, int n);
int emg3d_mg_vec_set(emg3d_mg_t* mg, int id, const void* host);
int emg3d_mg_vec_get(emg3d_mg_t* mg, int id, void* host);
int emg3d_mg_vec_copy(emg3d_mg_t* mg, int dst, int src);
int emg3d_mg_vec_axpy(emg3d_mg_t* mg, int y, double alpha_re, double alpha_im, int x);
int emg3d_mg_vec_scale(emg3d_mg_t* mg, int y, double alpha_re, double alpha_im);
int emg3d_mg_vec_dot(emg3d_mg_t* mg, int a, int b, double* out2);
int emg3d_mg_vec_amatvec(emg3d_mg_t* mg, int dst, int src);

#ifdef __cplusplus
}
#endif
#endif /* EMG3D_HIP_H */
