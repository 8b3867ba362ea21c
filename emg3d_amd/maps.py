"""Interpolation used either side of the multigrid path -- the interface of ``emg3d.maps.interp3d``
(reference emg3d/maps.py:179-276), evaluated on the device through the C ABI (``emg3d_interp3d``)."""
import numpy as np

from . import _lib


def interp3d(points, values, new_points, method, fill_value, mode, cval=0.0):
    """Interpolate ``values`` given on the regular grid ``points`` at ``new_points`` (reference
    emg3d/maps.py:179-276): ``method`` 'linear' (``RegularGridInterpolator``; ``fill_value=None``
    extrapolates) or 'cubic' (the SciPy spline arithmetic of the reference: not-a-knot index spline, cubic
    B-spline prefilter, 4x4x4 evaluation; points outside get ``cval``).  Fewer than four points along an
    axis force 'linear'.  Only ``mode='constant'`` is implemented on the device (what the receivers use);
    other modes raise ``NotImplementedError``."""
    if mode != 'constant':
        raise NotImplementedError("emg3d_amd.maps.interp3d: only mode='constant' runs on the device.")
    if method not in ('linear', 'cubic'):
        raise ValueError(f"`method` must be 'linear' or 'cubic'; provided: {method!r}.")
    lib = _lib.load()
    values = np.asarray(values)
    dtype = np.dtype(np.complex128 if np.iscomplexobj(values) else np.float64)
    pts = [np.ascontiguousarray(p, dtype=np.float64) for p in points]
    if values.shape != tuple(p.size for p in pts):
        raise ValueError(f"There are {tuple(p.size for p in pts)} points and {values.shape} values.")
    vals = np.ascontiguousarray(values.astype(dtype, copy=False).ravel(order='F'))
    xi = np.broadcast_arrays(*[np.asarray(c, dtype=np.float64) for c in new_points])
    shape = xi[0].shape
    n = int(xi[0].size)
    flat = np.ascontiguousarray(np.stack([c.ravel() for c in xi]))
    out = np.empty(max(n, 1), dtype=dtype)
    fill_c = None if fill_value is None else complex(np.asarray(fill_value).ravel()[0])
    if fill_c is not None and np.isnan(fill_c.real):
        fill_c = complex(np.nan, fill_c.imag)
    if n:
        _lib.check(lib.emg3d_interp3d(_lib.dtype_code(dtype), *(int(p.size) for p in pts), *(_lib.ptr(p) for p in pts),
                                      _lib.ptr(vals), n, _lib.ptr(flat), 0 if method == 'linear' else 1,
                                      0 if fill_c is None else 1, 0.0 if fill_c is None else fill_c.real,
                                      float(cval), _lib.ptr(out)), "emg3d_interp3d")
    if fill_c is not None and dtype.kind == 'c' and np.isnan(fill_c.real) and np.isnan(fill_c.imag):
        # a complex NaN fill value (0j * nan = nan + nan j, what fields.get_receiver passes): both parts
        bad = np.isnan(out.real)
        out[bad] = complex(np.nan, np.nan)
    return out[:n].reshape(shape)


def edges2cellaverages(ex, ey, ez, vol, out_x, out_y, out_z):
    """Interpolate fields defined on edges to volume-averaged cell values, ADDED into ``out_x/y/z`` in place --
    the interface of the reference's numba kernel ``maps.edges2cellaverages`` (emg3d/maps.py:578-630), evaluated on
    the device (``emg3d_edges2cellaverages``: one thread per cell, the reference's accumulation order)."""
    lib = _lib.load()
    dtype = np.dtype(np.complex128 if any(np.iscomplexobj(a) for a in (ex, ey, ez)) else np.float64)
    f = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=dtype).ravel(order='F') for a in (ex, ey, ez)]))
    nx, ny, nz = (int(v) for v in np.shape(vol))
    v = np.ascontiguousarray(np.asarray(vol, dtype=np.float64).ravel(order='F'))
    outs = [np.ascontiguousarray(np.asarray(o, dtype=dtype).ravel(order='F')) for o in (out_x, out_y, out_z)]
    _lib.check(lib.emg3d_edges2cellaverages(_lib.dtype_code(dtype), nx, ny, nz, _lib.ptr(f), _lib.ptr(v),
                                            *(_lib.ptr(o) for o in outs)), "emg3d_edges2cellaverages")
    for o, r in zip((out_x, out_y, out_z), outs):
        o[...] = r.reshape((nx, ny, nz), order='F') if np.iscomplexobj(o) or dtype == np.float64 else r.reshape((nx, ny, nz), order='F').real
