// Translation unit of the residual / operator kernels (stencil.hpp: k_residual, k_residual_zm): instantiations and launcher.
#include "sweep_launch.hpp"
#ifndef EMG3D_UNIT_T
#define EMG3D_UNIT_T 2
#endif

template <class T, int MODE>
static void residual_launch_m(int kz, dim3 grid, hipStream_t st, const ResidualArgs<T>& a) {
    if (kz == 4) hipLaunchKernelGGL((k_residual_zm<T, MODE, 4>), grid, dim3(EMG_BLOCK), 0, st, a);
    else if (kz == 8) hipLaunchKernelGGL((k_residual_zm<T, MODE, 8>), grid, dim3(EMG_BLOCK), 0, st, a);
#ifdef EMG3D_LAB
    else if (kz == 2) hipLaunchKernelGGL((k_residual_zm<T, MODE, 2>), grid, dim3(EMG_BLOCK), 0, st, a);
    else if (kz == 16) hipLaunchKernelGGL((k_residual_zm<T, MODE, 16>), grid, dim3(EMG_BLOCK), 0, st, a);
#endif
    else hipLaunchKernelGGL((k_residual<T, MODE>), grid, dim3(EMG_BLOCK), 0, st, a);
}
template <class T>
void residual_launch(int mode, int kz, dim3 grid, hipStream_t st, const ResidualArgs<T>& a) {
    if (mode == 0) hipLaunchKernelGGL((k_residual<T, 0>), grid, dim3(EMG_BLOCK), 0, st, a);
    else if (mode == 2) residual_launch_m<T, 2>(kz, grid, st, a);
    else residual_launch_m<T, 1>(kz, grid, st, a);
}
#if EMG3D_UNIT_T != 1
template void residual_launch<double>(int, int, dim3, hipStream_t, const ResidualArgs<double>&);
#endif
#if EMG3D_UNIT_T != 0
template void residual_launch<c128>(int, int, dim3, hipStream_t, const ResidualArgs<c128>&);
#endif
