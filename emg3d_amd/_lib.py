"""ctypes binding of ``libemg3d_hip.so`` (C ABI: ``include/emg3d_hip.h``).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, an exception is raised.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  EMG3D_HIP_LIB: another build of it (A/B runs in one gpurun call).
LIB_PATH = os.environ.get("EMG3D_HIP_LIB") or os.path.join(_HERE, "libemg3d_hip.so")
# The lab build (-DEMG3D_LAB): the same library plus the superseded kernel variants and one environment variable per
# tuning knob; only tests/test_gpu_variants.py and tools/ load it (`use(LAB_PATH)`).
LAB_PATH = os.path.join(_HERE, "libemg3d_hip_lab.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_double = ctypes.c_double
c_dp = ctypes.POINTER(ctypes.c_double)

# include/emg3d_hip.h: EMG3D_HIP_ABI_VERSION -- a library built from another header version is refused at load
ABI_VERSION = 103

# name -> (restype, argtypes); mirrors include/emg3d_hip.h one to one.
SIGNATURES = {
    "emg3d_hip_version": (c_int, []),
    "emg3d_hip_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "emg3d_hip_set_device": (c_int, [c_int]),
    "emg3d_hip_device_info": (c_int, [c_int, ctypes.c_char_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]),
    "emg3d_hip_mem_info": (c_int, [c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "emg3d_hip_release_cached": (c_i64, []),
    "emg3d_hip_cached_bytes": (c_i64, []),
    "emg3d_hip_cached_bytes_on": (c_i64, [c_int]),
    "emg3d_mg_placement": (c_int, [c_vp, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_float)]),
    "emg3d_amat_x": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "emg3d_get_h_field": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_double, c_double]),
    "emg3d_gauss_seidel": (c_int, [c_int, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_int, c_int]),
    "emg3d_restrict": (c_int, [c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_int]),
    "emg3d_restrict_weights": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "emg3d_solve": (c_int, [c_int, c_vp, c_vp, c_i64]),
    "emg3d_blocks_to_amat": (c_int, [c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64]),
    "emg3d_prolongation": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "emg3d_restrict_model": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_int]),
    "emg3d_sweep_plan": (c_int, [c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "emg3d_mg_create": (c_int, [ctypes.POINTER(c_vp), c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp, c_int]),
    "emg3d_mg_destroy": (None, [c_vp]),
    "emg3d_mg_create_sv": (c_int, [ctypes.POINTER(c_vp), c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_vp, c_double, c_double, c_int]),
    "emg3d_mg_create_vs": (c_int, [ctypes.POINTER(c_vp), c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_vp, c_vp, c_double, c_double, c_int, c_int]),
    "emg3d_mg_create_vse": (c_int, [ctypes.POINTER(c_vp), c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_double, c_double, c_double, c_int, c_int]),
    "emg3d_mg_set_smu0_eps": (c_int, [c_vp, c_double, c_double, c_double]),
    "emg3d_mg_set_sfield_vector": (c_int, [c_vp, c_vp, c_double, c_double]),
    "emg3d_mg_set_sfield_dipole": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "emg3d_source_field": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "emg3d_mg_prepare": (c_int, [c_vp, c_int, c_int]),
    "emg3d_mg_set_trace": (c_int, [c_vp, c_int]),
    "emg3d_mg_get_trace": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp]),
    "emg3d_mg_set_params": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int]),
    "emg3d_mg_set_sfield": (c_int, [c_vp, c_vp]),
    "emg3d_mg_set_efield": (c_int, [c_vp, c_vp]),
    "emg3d_mg_get_efield": (c_int, [c_vp, c_vp]),
    "emg3d_mg_get_residual": (c_int, [c_vp, c_vp]),
    "emg3d_mg_get_hfield": (c_int, [c_vp, c_int, c_double, c_double, c_vp]),
    "emg3d_interp3d": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_double,
                               c_double, c_vp]),
    "emg3d_get_receiver_response": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64,
                                            c_vp, c_vp, c_vp]),
    "emg3d_mg_get_receiver_response": (c_int, [c_vp, c_int, c_int, c_double, c_double, c_i64, c_vp, c_vp, c_vp]),
    "emg3d_edges2cellaverages": (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "emg3d_mg_gradient": (c_int, [c_vp, c_int, c_double, c_double, c_vp]),
    "emg3d_mg_set_batch": (c_int, [c_vp, c_int]),
    "emg3d_mg_get_batch": (c_int, [c_vp]),
    "emg3d_mg_select": (c_int, [c_vp, c_int]),
    "emg3d_mg_set_mask": (c_int, [c_vp, c_vp]),
    "emg3d_mg_residual_norm": (c_int, [c_vp, c_dp]),
    "emg3d_mg_sfield_norm": (c_int, [c_vp, c_dp]),
    "emg3d_mg_smooth": (c_int, [c_vp, c_int, c_int]),
    "emg3d_mg_begin": (c_int, [c_vp, c_int]),
    "emg3d_mg_set_smu0": (c_int, [c_vp, ctypes.c_double, ctypes.c_double]),
    "emg3d_mg_cycle": (c_int, [c_vp, c_int, c_int, c_dp]),
    "emg3d_mg_cycle_next": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_dp]),
    "emg3d_mg_cycles": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "emg3d_mg_efield_devptr": (c_vp, [c_vp]),
    "emg3d_mg_sfield_devptr": (c_vp, [c_vp]),
    "emg3d_mg_stream": (c_vp, [c_vp]),
    "emg3d_mg_nE": (c_i64, [c_vp]),
    "emg3d_mg_sync": (c_int, [c_vp]),
    "emg3d_mg_device_bytes": (c_i64, [c_vp]),
    "emg3d_mg_time_sweep": (c_int, [c_vp, c_int, c_int, ctypes.POINTER(ctypes.c_float)]),
    "emg3d_mg_last_sweep_kernel": (c_int, [c_vp, ctypes.c_char_p]),
    "emg3d_mg_time_residual": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_float)]),
    "emg3d_mg_last_residual_kernel": (c_int, [c_vp, ctypes.c_char_p]),
    "emg3d_mg_amatvec": (c_int, [c_vp, c_vp, c_vp]),
    "emg3d_mg_vec_alloc": (c_int, [c_vp, c_int]),
    "emg3d_mg_vec_set": (c_int, [c_vp, c_int, c_vp]),
    "emg3d_mg_vec_get": (c_int, [c_vp, c_int, c_vp]),
    "emg3d_mg_vec_copy": (c_int, [c_vp, c_int, c_int]),
    "emg3d_mg_vec_axpy": (c_int, [c_vp, c_int, c_double, c_double, c_int]),
    "emg3d_mg_vec_scale": (c_int, [c_vp, c_int, c_double, c_double]),
    "emg3d_mg_vec_dot": (c_int, [c_vp, c_int, c_int, c_dp]),
    "emg3d_mg_vec_amatvec": (c_int, [c_vp, c_int, c_int]),
}

_lib = None
_path = None
_loaded = {}


class HipLibraryError(RuntimeError):
    """The HIP extension is missing or a device call failed."""


def _open(path):
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found: the HIP extension is not built. Build it with\n"
            "  python -c 'import __graft_entry__ as g; g.build()'   (needs hipcc)\n"
            "emg3d_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    try:
        lib.emg3d_hip_version.restype = c_int
        got = int(lib.emg3d_hip_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise HipLibraryError(
            f"{path} has ABI version {got}, this package needs {ABI_VERSION} (include/emg3d_hip.h: EMG3D_HIP_ABI_VERSION): "
            "a stale build. Rebuild it with\n  python -c 'import __graft_entry__ as g; g.build(force=True)'")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _loaded[path] = lib
    return lib


def load():
    """The library of this process (loaded once; every prototype declared)."""
    global _lib, _path
    if _lib is None:
        _lib, _path = _open(LIB_PATH), LIB_PATH
    return _lib


def use(path=None):
    """Make another build of the library the one `load()` returns (None: back to LIB_PATH).  Objects created before
    keep the build they were created with.  Returns the previous path."""
    global _lib, _path
    prev = _path or LIB_PATH
    want = path or LIB_PATH
    _lib, _path = _open(want), want
    return prev


def check(status, what):
    if status != 0:
        raise HipLibraryError(f"{what} failed with status {status} "
                              f"({'invalid argument' if status < 0 else 'HIP error'})")


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.complex128:
        return 1
    if dtype == np.float64:
        return 0
    raise TypeError(f"fields must be float64 or complex128, got {dtype}")


def ptr(a):
    return a.ctypes.data_as(c_vp)


def device_count():
    n = c_int(0)
    check(load().emg3d_hip_device_count(ctypes.byref(n)), "emg3d_hip_device_count")
    return n.value


def device_info(device=0):
    name = ctypes.create_string_buffer(256)
    mem = c_i64(0)
    cus = c_int(0)
    check(load().emg3d_hip_device_info(device, name, ctypes.byref(mem), ctypes.byref(cus)),
          "emg3d_hip_device_info")
    return {"name": name.value.decode(), "total_mem": mem.value, "cu_count": cus.value}


def mem_info(device=0):
    """(free, total) bytes of `device` as the driver reports them now, and what this process's own block pool holds (parked
    blocks count as used for the driver, but the next allocation of the process may take them)."""
    fr, tot = c_i64(0), c_i64(0)
    check(load().emg3d_hip_mem_info(device, ctypes.byref(fr), ctypes.byref(tot)), "emg3d_hip_mem_info")
    return {"free": fr.value, "total": tot.value, "pooled": int(load().emg3d_hip_cached_bytes()),
            "pooled_on_device": int(load().emg3d_hip_cached_bytes_on(device))}


def sweep_plan(vnC, direction, dtype=np.complex128, ordering='colour', nsys=1, cu_count=0):
    """The line-sweep kernel the library selects for a level of ``vnC`` cells along ``direction`` (1, 2, 3) on a device of
    ``cu_count`` compute units (0: the current device) -- ``emg3d_sweep_plan``: shape logic only, runs without a GPU when
    ``cu_count`` > 0.  Returns the instantiation's name and the launch shape."""
    name = ctypes.create_string_buffer(64)
    info = (c_i64 * 6)()
    check(load().emg3d_sweep_plan(dtype_code(dtype), int(vnC[0]), int(vnC[1]), int(vnC[2]), int(direction),
                                  1 if ordering == 'colour' else 0, int(nsys), int(cu_count), name, info), "emg3d_sweep_plan")
    return {"kernel": name.value.decode(), "lines_per_colour": info[0], "lines_per_wave": info[1], "rounds": info[2],
            "factor_kind": info[3], "split": bool(info[4]), "big_offsets": bool(info[5])}
