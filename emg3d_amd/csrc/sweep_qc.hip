// Translation unit of the quad-per-line chain kernel (smooth_qc.hpp): its instantiations and their launcher (sweep_launch.hpp).
#include "sweep_launch.hpp"
// -DEMG3D_UNIT_T=0 | 1: only the float64 | complex128 instantiations (the build compiles the heavy families once per type)
#ifndef EMG3D_UNIT_T
#define EMG3D_UNIT_T 2
#endif
#include "smooth_qc.hpp"

template <class T, int ST, int LPW, bool BIG>
static void qc_launch_z(bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (zsep) hipLaunchKernelGGL((k_line_sweep_qc<T, ST, LPW, true, BIG>), grid, dim3(EMG_Q_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_line_sweep_qc<T, ST, LPW, false, BIG>), grid, dim3(EMG_Q_BLOCK), 0, st, a);
}
template <class T, int ST>
static void qc_launch_s(int lpw, bool zsep, bool big, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (big) qc_launch_z<T, ST, 16, true>(zsep, grid, st, a);
    else if (lpw == 16) qc_launch_z<T, ST, 16, false>(zsep, grid, st, a);
    else if (lpw == 8) qc_launch_z<T, ST, 8, false>(zsep, grid, st, a);
    else if (lpw == 2) qc_launch_z<T, ST, 2, false>(zsep, grid, st, a);
    else qc_launch_z<T, ST, 4, false>(zsep, grid, st, a);
}
template <class T>
void qc_launch(int stages, int lpw, bool zsep, bool big, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (stages == 2) qc_launch_s<T, 2>(lpw, zsep, big, grid, st, a);
    else qc_launch_s<T, 3>(lpw, zsep, big, grid, st, a);
}
#if EMG3D_UNIT_T != 1
template void qc_launch<double>(int, int, bool, bool, dim3, hipStream_t, const LineArgs<double>&);
#endif
#if EMG3D_UNIT_T != 0
template void qc_launch<c128>(int, int, bool, bool, dim3, hipStream_t, const LineArgs<c128>&);
#endif
