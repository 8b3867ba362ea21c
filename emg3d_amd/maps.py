"""Interpolation used either side of the multigrid path -- the interface of ``emg3d.maps.interp3d``
(reference emg3d/maps.py:179-276), evaluated on the device through the C ABI (``emg3d_interp3d``)."""
import numpy as np

from . import _lib


_NPAD = 12      # scipy.ndimage pads 'nearest' inputs by 12 samples before the spline filter (_prepad_for_spline_filter)


def _index_coords(points, xi):
    """Coordinates -> index coordinates of the grid vectors, as the reference does it (maps.py:255-260: not-a-knot
    cubic interpolation of the index, extrapolated)."""
    from scipy import interpolate
    return [interpolate.interp1d(p, np.arange(p.size), kind='cubic', bounds_error=False,
                                 fill_value='extrapolate')(c) for p, c in zip(points, xi)]


def _wrap_coords(c, n):
    """scipy.ndimage's map_coordinate for mode 'wrap' (ni_interpolation.c, NI_EXTEND_WRAP): period n - 1, the integer part
    of the quotient truncated towards zero."""
    c = np.array(c, dtype=np.float64)
    sz = float(n - 1)
    lo = c < 0
    hi = c > n - 1
    c[lo] = c[lo] + sz * (np.trunc(-c[lo] / sz) + 1.0)
    c[hi] = c[hi] - sz * np.trunc(c[hi] / sz)
    return c


def interp3d(points, values, new_points, method, fill_value, mode, cval=0.0):
    """Interpolate ``values`` given on the regular grid ``points`` at ``new_points`` (reference
    emg3d/maps.py:179-276): ``method`` 'linear' (``RegularGridInterpolator``; ``fill_value=None``
    extrapolates) or 'cubic' (the SciPy spline arithmetic of the reference: not-a-knot index spline, cubic
    B-spline prefilter, 4x4x4 evaluation).  Fewer than four points along an axis force 'linear'.

    ``mode`` (cubic only, ``scipy.ndimage.map_coordinates``): 'constant' (points outside get ``cval``), 'nearest' (what
    ``fields.get_receiver(extrapolate=True)`` uses: the array is extended by its edge values), 'mirror', 'reflect' and 'wrap'
    (SciPy's legacy rule: coordinates wrapped with period n - 1, mirror spline) -- the prefilter and the evaluation over the
    whole array run in HBM, the boundary rule is applied to the O(n_points) index coordinates on the host."""
    if mode not in ('constant', 'nearest', 'mirror', 'reflect', 'wrap'):
        raise ValueError(f"emg3d_amd.maps.interp3d: unknown mode {mode!r} "
                         "('constant', 'nearest', 'mirror', 'reflect', 'wrap').")
    if method not in ('linear', 'cubic'):
        raise ValueError(f"`method` must be 'linear' or 'cubic'; provided: {method!r}.")
    lib = _lib.load()
    values = np.asarray(values)
    dtype = np.dtype(np.complex128 if np.iscomplexobj(values) else np.float64)
    pts = [np.ascontiguousarray(p, dtype=np.float64) for p in points]
    if values.shape != tuple(p.size for p in pts):
        raise ValueError(f"There are {tuple(p.size for p in pts)} points and {values.shape} values.")
    xi = np.broadcast_arrays(*[np.asarray(c, dtype=np.float64) for c in new_points])
    shape = xi[0].shape
    n = int(xi[0].size)
    code = 0 if method == 'linear' else 1
    if code == 1 and mode != 'constant' and all(p.size >= 4 for p in pts):
        # boundary modes of map_coordinates: index coordinates on the host (O(n) work), then the cubic spline of the
        # (edge-padded) array on the device on INDEX coordinates (method codes 2, 3)
        co = _index_coords(pts, [c.ravel() for c in xi])
        if mode == 'nearest':       # edge-padded array, stencil at the shifted coordinate, indices clamped (code 3)
            values = np.pad(values, _NPAD, mode='edge')
            co = [c + _NPAD for c in co]
            code = 3
        elif mode == 'reflect':     # stencil at the coordinate, indices reflected, "reflect" prefilter (code 4)
            code = 4
        else:                       # stencil at the coordinate, indices mirrored (code 2)
            if mode == 'wrap':      # scipy's NI_EXTEND_WRAP: coordinates wrapped with period n - 1, then the mirror spline
                co = [_wrap_coords(c, m) for c, m in zip(co, values.shape)]
            code = 2
        pts = [np.arange(m, dtype=np.float64) for m in values.shape]
        xi = co
    vals = np.ascontiguousarray(values.astype(dtype, copy=False).ravel(order='F'))
    flat = np.ascontiguousarray(np.stack([np.asarray(c).ravel() for c in xi]))
    out = np.empty(max(n, 1), dtype=dtype)
    fill_c = None if fill_value is None else complex(np.asarray(fill_value).ravel()[0])
    if fill_c is not None and np.isnan(fill_c.real):
        fill_c = complex(np.nan, fill_c.imag)
    if n:
        _lib.check(lib.emg3d_interp3d(_lib.dtype_code(dtype), *(int(p.size) for p in pts), *(_lib.ptr(p) for p in pts),
                                      _lib.ptr(vals), n, _lib.ptr(flat), code,
                                      0 if fill_c is None else 1, 0.0 if fill_c is None else fill_c.real,
                                      float(cval), _lib.ptr(out)), "emg3d_interp3d")
    if code < 2 and fill_c is not None and dtype.kind == 'c' and np.isnan(fill_c.real) and np.isnan(fill_c.imag):
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
