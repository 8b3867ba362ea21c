// Host-side entry points of the four line-sweep kernel families (reference emg3d/core.py:477-1316, the line solves).
//
// Each family is compiled in a translation unit of its own -- sweep_qc.hip, sweep_thm.hip, sweep_tha.hip, sweep_qpl.hip: the
// kernels of smooth_qc.hpp / _thm / _tha / _qpl instantiated for float64 and complex128 -- so that the library builds in parallel
// (one unit: 100 s; five: the longest of them).  The cycle driver (mg.hpp) chooses the instantiation by runtime values and never
// names a kernel of these families itself; a value without an instantiation falls to the nearest one as noted.
#pragma once
#include "smooth.hpp"
#include "stencil.hpp"

// k_line_sweep_qc<T, stages, lpw, zsep, big>: stages 2 | 3 (else 3); lines per wave 16 | 8 | 2 (else 4); big: 64-bit field offsets
// (the 16-line instantiation only).  Workgroups of EMG_Q_BLOCK threads.
template <class T> void qc_launch(int stages, int lpw, bool zsep, bool big, dim3 grid, hipStream_t st, const LineArgs<T>& a);

// k_line_sweep_thm<T, stages, lpw, KL, zsep>: stages 3 | 2 (else 3); lines per pair of waves 4 | 12 (else 8); lifo (lab build only):
// the last KL forward steps of a half kept in dynamic LDS.  Workgroups of EMG_RP_BLOCK threads.
template <class T> void thm_launch(int stages, int lpw, bool lifo, bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a);
// lab build: asks for the LIFO instantiations' dynamic LDS (> 64 KB must be asked for per instantiation and device)
template <class T> void thm_lifo_attrs();

// k_line_sweep_tha<T, helpers, zsep>: helper waves per half 3 (lab build: 2).  tha_attrs asks for max_dyn_lds bytes of dynamic LDS
// for every instantiation on the current device; false: refused (the caller selects another kernel).
template <class T> void tha_launch(int helpers, bool zsep, dim3 grid, size_t dyn_lds, hipStream_t st, const LineArgs<T>& a);
template <class T> bool tha_attrs(int max_dyn_lds);

// k_line_sweep_qpl<T, nw, m, hl, dm>: waves per workgroup 1 | 2 | 4 (else 8), blocks per quad 1 | 2; hl: the hyperplane loop of the
// lexicographic order (m = 1); dm: launch descriptors 0 none | 1 generate | 2 load (nw = 1, colour order); chain: the chain form of
// the two recurrences instead of the scans (nw = 1, m = 1, lines of <= 8 blocks, dm 0 | 2).
template <class T> void qpl_launch(int nw, int m, bool hl, int dm, bool chain, dim3 grid, hipStream_t st, const LineArgs<T>& a);

// k_residual<T, mode> (kz <= 1) / k_residual_zm<T, mode, kz> (stencil.hpp; reference core.amat_x, emg3d/core.py:29-177, and
// solver.residual, solver.py:980-1039): mode 0 = r -= A e (the operator itself), 1 = r = s - A e, 2 = the norm's partial sums only;
// kz node planes per thread 4 | 8 (lab build: also 2 | 16; modes 1, 2).  Compiled in stencil_res.hip.  Workgroups of EMG_BLOCK threads.
template <class T> void residual_launch(int mode, int kz, dim3 grid, hipStream_t st, const ResidualArgs<T>& a);
