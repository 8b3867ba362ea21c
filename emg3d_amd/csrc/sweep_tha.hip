// Translation unit of the two-sided affine kernel with helper waves (smooth_tha.hpp): instantiations and launcher.
#include "sweep_launch.hpp"
// -DEMG3D_UNIT_T=0 | 1: only the float64 | complex128 instantiations (the build compiles the heavy families once per type)
#ifndef EMG3D_UNIT_T
#define EMG3D_UNIT_T 2
#endif
#include "smooth_tha.hpp"

template <class T, int NH>
static void tha_launch_h(bool zsep, dim3 grid, size_t dyn, hipStream_t st, const LineArgs<T>& a) {
    if (zsep) hipLaunchKernelGGL((k_line_sweep_tha<T, NH, true>), grid, dim3(tha_threads<NH>()), dyn, st, a);
    else hipLaunchKernelGGL((k_line_sweep_tha<T, NH, false>), grid, dim3(tha_threads<NH>()), dyn, st, a);
}
template <class T>
void tha_launch(int helpers, bool zsep, dim3 grid, size_t dyn_lds, hipStream_t st, const LineArgs<T>& a) {
#ifdef EMG3D_LAB
    if (helpers == 2) { tha_launch_h<T, 2>(zsep, grid, dyn_lds, st, a); return; }
#endif
    (void)helpers;
    tha_launch_h<T, 3>(zsep, grid, dyn_lds, st, a);
}
#if EMG3D_UNIT_T != 1
template void tha_launch<double>(int, bool, dim3, size_t, hipStream_t, const LineArgs<double>&);
#endif
#if EMG3D_UNIT_T != 0
template void tha_launch<c128>(int, bool, dim3, size_t, hipStream_t, const LineArgs<c128>&);
#endif

template <class T>
bool tha_attrs(int max_dyn_lds) {
    bool ok = true;
    auto dyn_lds = [&](const void* f) {
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn_lds) != hipSuccess) { (void)hipGetLastError(); ok = false; }
    };
    dyn_lds(reinterpret_cast<const void*>(&k_line_sweep_tha<T, 3, false>));
    dyn_lds(reinterpret_cast<const void*>(&k_line_sweep_tha<T, 3, true>));
#ifdef EMG3D_LAB
    dyn_lds(reinterpret_cast<const void*>(&k_line_sweep_tha<T, 2, false>));
    dyn_lds(reinterpret_cast<const void*>(&k_line_sweep_tha<T, 2, true>));
#endif
    return ok;
}
#if EMG3D_UNIT_T != 1
template bool tha_attrs<double>(int);
#endif
#if EMG3D_UNIT_T != 0
template bool tha_attrs<c128>(int);
#endif
