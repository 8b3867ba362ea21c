// Translation unit of the two-sided chain kernel on the mirrored factorisation (smooth_thm.hpp): instantiations and launcher.
#include "sweep_launch.hpp"
// -DEMG3D_UNIT_T=0 | 1: only the float64 | complex128 instantiations (the build compiles the heavy families once per type)
#ifndef EMG3D_UNIT_T
#define EMG3D_UNIT_T 2
#endif
#include "smooth_thm.hpp"

template <class T, int ST, int LPW, int KL>
static void thm_launch_k(bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    constexpr size_t dyn = thm_lifo_bytes<T, LPW, KL>();
    if (zsep) hipLaunchKernelGGL((k_line_sweep_thm<T, ST, LPW, KL, true>), grid, dim3(EMG_RP_BLOCK), dyn, st, a);
    else hipLaunchKernelGGL((k_line_sweep_thm<T, ST, LPW, KL, false>), grid, dim3(EMG_RP_BLOCK), dyn, st, a);
}
template <class T, int ST, int LPW>
static void thm_launch_l(bool lifo, bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
#ifdef EMG3D_LAB
    if (lifo) { thm_launch_k<T, ST, LPW, (LPW == 12) ? 10 : 15>(zsep, grid, st, a); return; }
#endif
    (void)lifo;
    thm_launch_k<T, ST, LPW, 0>(zsep, grid, st, a);
}
template <class T, int ST>
static void thm_launch_s(int lpw, bool lifo, bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (lpw == 4) thm_launch_l<T, ST, 4>(lifo, zsep, grid, st, a);
    else if (lpw == 12) thm_launch_l<T, ST, 12>(lifo, zsep, grid, st, a);
    else thm_launch_l<T, ST, 8>(lifo, zsep, grid, st, a);
}
template <class T>
void thm_launch(int stages, int lpw, bool lifo, bool zsep, dim3 grid, hipStream_t st, const LineArgs<T>& a) {
    if (stages == 2) thm_launch_s<T, 2>(lpw, lifo, zsep, grid, st, a);
    else thm_launch_s<T, 3>(lpw, lifo, zsep, grid, st, a);
}
#if EMG3D_UNIT_T != 1
template void thm_launch<double>(int, int, bool, bool, dim3, hipStream_t, const LineArgs<double>&);
#endif
#if EMG3D_UNIT_T != 0
template void thm_launch<c128>(int, int, bool, bool, dim3, hipStream_t, const LineArgs<c128>&);
#endif

#ifdef EMG3D_LAB
template <class T, int ST, int LPW, int KL>
static void thm_attr() {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_line_sweep_thm<T, ST, LPW, KL, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)thm_lifo_bytes<T, LPW, KL>()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_line_sweep_thm<T, ST, LPW, KL, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)thm_lifo_bytes<T, LPW, KL>()) != hipSuccess)
        (void)hipGetLastError();
}
template <class T>
void thm_lifo_attrs() {
    thm_attr<T, 3, 4, 15>(); thm_attr<T, 3, 8, 15>(); thm_attr<T, 3, 12, 10>();
    thm_attr<T, 2, 4, 15>(); thm_attr<T, 2, 8, 15>(); thm_attr<T, 2, 12, 10>();
}
#else
template <class T> void thm_lifo_attrs() {}
#endif
#if EMG3D_UNIT_T != 1
template void thm_lifo_attrs<double>();
#endif
#if EMG3D_UNIT_T != 0
template void thm_lifo_attrs<c128>();
#endif
