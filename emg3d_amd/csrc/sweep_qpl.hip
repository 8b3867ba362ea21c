// Translation unit of the quad-per-block scan kernel (smooth_qpl.hpp): instantiations and launcher.
#include "sweep_launch.hpp"
// -DEMG3D_UNIT_T=0 | 1: only the float64 | complex128 instantiations (the build compiles the heavy families once per type)
#ifndef EMG3D_UNIT_T
#define EMG3D_UNIT_T 2
#endif
#include "smooth_qpl.hpp"

template <class T, int NW, int M>
static void qpl_launch_c(dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    hipLaunchKernelGGL((k_line_sweep_qpl<T, NW, M>), grid, dim3(64 * NW), 0, st, a);
}
template <class T, int NW>
static void qpl_launch_hl(dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    hipLaunchKernelGGL((k_line_sweep_qpl<T, NW, 1, true>), grid, dim3(64 * NW), 0, st, a);
}
template <class T, int M>
static void qpl_launch_m(int nw, int dm, bool chain, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if constexpr (M == 1) {
        if (chain && dm != 1 && nw == 1) {
            if (dm == 2) hipLaunchKernelGGL((k_line_sweep_qpl<T, 1, 1, false, 2, true>), grid, dim3(64), 0, st, a);
            else hipLaunchKernelGGL((k_line_sweep_qpl<T, 1, 1, false, 0, true>), grid, dim3(64), 0, st, a);
            return;
        }
    }
    if (dm == 1) hipLaunchKernelGGL((k_line_sweep_qpl<T, 1, M, false, 1>), grid, dim3(64), 0, st, a);
    else if (dm == 2) hipLaunchKernelGGL((k_line_sweep_qpl<T, 1, M, false, 2>), grid, dim3(64), 0, st, a);
    else if (nw == 1) qpl_launch_c<T, 1, M>(grid, st, a);
    else if (nw == 2) qpl_launch_c<T, 2, M>(grid, st, a);
    else if (nw == 4) qpl_launch_c<T, 4, M>(grid, st, a);
    else qpl_launch_c<T, 8, M>(grid, st, a);
}
template <class T>
void qpl_launch(int nw, int m, bool hl, int dm, bool chain, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (hl) {
        if (nw == 1) qpl_launch_hl<T, 1>(grid, st, a);
        else if (nw == 2) qpl_launch_hl<T, 2>(grid, st, a);
        else if (nw == 4) qpl_launch_hl<T, 4>(grid, st, a);
        else qpl_launch_hl<T, 8>(grid, st, a);
    } else if (m == 2) qpl_launch_m<T, 2>(nw, dm, false, grid, st, a);
    else qpl_launch_m<T, 1>(nw, dm, chain, grid, st, a);
}
#if EMG3D_UNIT_T != 1
template void qpl_launch<double>(int, int, bool, int, bool, dim3, hipStream_t, const LineArgs<double>&);
#endif
#if EMG3D_UNIT_T != 0
template void qpl_launch<c128>(int, int, bool, int, bool, dim3, hipStream_t, const LineArgs<c128>&);
#endif
