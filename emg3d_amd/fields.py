"""Field containers on the boundary of the hot path: ``Field`` / ``SourceField``
(reference emg3d/fields.py:34-443) and a point/finite-dipole source builder
(the part of ``get_source_field`` the benchmark configurations need,
fields.py:446-642, 914-1010)."""
import warnings

import numpy as np
from scipy.constants import mu_0

__all__ = ['Field', 'SourceField', 'get_source_field', 'get_h_field']


class Field(np.ndarray):
    """1-D array ``[fx, fy, fz]`` with F-ordered component views.

    ``Field(grid, dtype=complex)`` -> zeros; ``Field(grid, array)`` -> wraps the
    array (no copy); ``Field(fx, fy, fz)`` -> concatenates three 3-D arrays.
    ``freq`` > 0: frequency domain (s = -2 i pi f); < 0: Laplace domain (s = f).
    """

    def __new__(cls, fx_or_grid, fy_or_field=None, fz=None, dtype=np.complex128, freq=None):
        if fz is None and getattr(fx_or_grid, 'nEx', 1) is None:           # (reference fields.py:128-130)
            raise ValueError("Provided grid must be a 3D grid.")
        if fy_or_field is None and fz is None:
            data = np.zeros(fx_or_grid.nE, dtype=dtype)
            shapes = (fx_or_grid.vnEx, fx_or_grid.vnEy, fx_or_grid.vnEz)
        elif fz is None:
            data = fy_or_field
            shapes = (fx_or_grid.vnEx, fx_or_grid.vnEy, fx_or_grid.vnEz)
        else:
            data = np.r_[fx_or_grid.ravel('F'), fy_or_field.ravel('F'), fz.ravel('F')]
            shapes = (fx_or_grid.shape, fy_or_field.shape, fz.shape)
        obj = np.asarray(data).view(cls)
        obj.vnEx, obj.vnEy, obj.vnEz = (tuple(int(n) for n in s) for s in shapes)
        obj.nEx, obj.nEy, obj.nEz = (int(np.prod(s)) for s in shapes)
        if freq is None and hasattr(fy_or_field, '_freq'):
            freq = fy_or_field._freq
        if freq == 0.0:
            raise ValueError("`freq` must be >0 (frequency domain) or <0 (Laplace domain).\n"
                             f"Provided frequency: {freq} Hz.")
        obj._freq = freq
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for name in ('nEx', 'nEy', 'nEz', 'vnEx', 'vnEy', 'vnEz', '_freq'):
            setattr(self, name, getattr(obj, name, None))

    def __reduce__(self):
        state = super().__reduce__()
        extra = tuple(getattr(self, n) for n in ('nEx', 'nEy', 'nEz', 'vnEx', 'vnEy', 'vnEz', '_freq'))
        return (state[0], state[1], state[2] + extra)

    def __setstate__(self, state):
        names = ('nEx', 'nEy', 'nEz', 'vnEx', 'vnEy', 'vnEz', '_freq')
        for name, value in zip(names, state[-len(names):]):
            setattr(self, name, value)
        super().__setstate__(state[:-len(names)])

    def amp(self):
        """Amplitude (reference fields.py:283-285)."""
        return EMArray(self.view()).amp()

    def pha(self, deg=False, unwrap=True, lag=True):
        """Phase (reference fields.py:287-305)."""
        return EMArray(self.view()).pha(deg, unwrap, lag)

    def to_dict(self, copy=False):
        """The information needed to rebuild the field (reference fields.py: Field.to_dict)."""
        from copy import deepcopy
        out = {'field': np.array(self.field), 'freq': self._freq, 'vnEx': self.vnEx, 'vnEy': self.vnEy, 'vnEz': self.vnEz,
               '__class__': self.__class__.__name__}
        return deepcopy(out) if copy else out

    @classmethod
    def from_dict(cls, inp):
        """Inverse of ``to_dict``; needs the keys field, freq, vnEx, vnEy, vnEz."""
        class Grid:
            pass
        grid = Grid()
        try:
            field, freq = inp['field'], inp['freq']
            grid.vnEx, grid.vnEy, grid.vnEz = inp['vnEx'], inp['vnEy'], inp['vnEz']
        except KeyError as e:
            raise KeyError(f"Variable {e} missing in `inp`.") from e
        grid.nEx, grid.nEy, grid.nEz = (int(np.prod(v)) for v in (grid.vnEx, grid.vnEy, grid.vnEz))
        grid.nE = grid.nEx + grid.nEy + grid.nEz
        return cls(grid, field, freq=freq)

    def copy(self):
        out = np.array(self).view(type(self))
        out.__array_finalize__(self)
        return out

    @property
    def field(self):
        return self.view()

    @field.setter
    def field(self, value):
        self.view()[:] = value

    @property
    def fx(self):
        return self.view()[:self.nEx].reshape(self.vnEx, order='F')

    @fx.setter
    def fx(self, value):
        self.view()[:self.nEx] = value.ravel('F')

    @property
    def fy(self):
        return self.view()[self.nEx:self.nEx + self.nEy].reshape(self.vnEy, order='F')

    @fy.setter
    def fy(self, value):
        self.view()[self.nEx:self.nEx + self.nEy] = value.ravel('F')

    @property
    def fz(self):
        return self.view()[self.nEx + self.nEy:].reshape(self.vnEz, order='F')

    @fz.setter
    def fz(self, value):
        self.view()[self.nEx + self.nEy:] = value.ravel('F')

    @property
    def freq(self):
        return None if self._freq is None else abs(self._freq)

    @property
    def sval(self):
        """s = -2 i pi f (frequency domain) or s = f (Laplace domain)."""
        if self._freq is None:
            return None
        return np.array(self._freq) if self._freq < 0 else np.array(-2j * np.pi * self._freq)

    @property
    def smu0(self):
        """s * mu_0 (mu_0 from scipy.constants, as the reference does)."""
        return None if self._freq is None else self.sval * mu_0

    @property
    def ensure_pec(self):
        """Zero the tangential components on the six boundary faces."""
        fx, fy, fz = self.fx, self.fy, self.fz
        fx[:, [0, -1], :] = 0.
        fx[:, :, [0, -1]] = 0.
        fy[[0, -1], :, :] = 0.
        fy[:, :, [0, -1]] = 0.
        fz[[0, -1], :, :] = 0.
        fz[:, [0, -1], :] = 0.

    @property
    def is_electric(self):
        return self.vnEx[0] < self.vnEy[0]


class SourceField(Field):
    """Field that requires ``freq``; dtype follows the domain."""

    def __new__(cls, fx_or_grid, fy_or_field=None, fz=None, dtype=np.complex128, freq=None):
        if freq is None:
            raise ValueError("SourceField requires the frequency.")
        dtype = complex if freq > 0 else float
        return super().__new__(cls, fx_or_grid, fy_or_field=fy_or_field, fz=fz, dtype=dtype, freq=freq)

    @property
    def vector(self):
        return np.real(self.field / self.smu0)

    @property
    def vx(self):
        return np.real(self.field.fx / self.smu0)

    @property
    def vy(self):
        return np.real(self.field.fy / self.smu0)

    @property
    def vz(self):
        return np.real(self.field.fz / self.smu0)


class FrequencySpec:
    """What a solve needs of a ``SourceField`` when the source itself is built in HBM (``solve(source=...)``,
    ``solve_sources``): the frequency, ``s``, ``s mu_0`` and the dtype -- without an nE-sized host array behind it
    (812 MB at 256^3)."""

    def __init__(self, freq):
        if freq is None or freq == 0.0:
            raise ValueError("`freq` must be >0 (frequency domain) or <0 (Laplace domain).\n"
                             f"Provided frequency: {freq} Hz.")
        self._freq = float(freq)
        self.dtype = np.dtype(np.complex128 if freq > 0 else np.float64)

    @property
    def freq(self):
        return abs(self._freq)

    @property
    def sval(self):
        return np.array(self._freq) if self._freq < 0 else np.array(-2j * np.pi * self._freq)

    @property
    def smu0(self):
        return self.sval * mu_0


def _dipole_from_point(src, length):
    """[x, y, z, azimuth, dip] -> [x0, x1, y0, y1, z0, z1] of given length (reference
    ``_finite_dipole_from_point_dipole``, emg3d/fields.py:1037-1040: same ``cosdg`` / ``sindg`` rotation)."""
    factors = _rotation(src[3], src[4]) * length / 2
    return np.ravel(src[:3] + np.stack([-factors, factors]), 'F')


def _loop_from_point(src, length):
    """Magnetic point dipole [x, y, z, azimuth, dip] -> the five corner points (3, 5) of a closed square loop of side
    ``length`` perpendicular to it (reference ``_square_loop_from_point_dipole``, emg3d/fields.py:1043-1049)."""
    half_diagonal = np.sqrt(2) * length / 2
    hor = _rotation(src[3] + 90, 0) * half_diagonal
    ver = _rotation(src[3], src[4] + 90) * half_diagonal
    return (src[:3] + np.stack([hor, ver, -hor, -ver, hor])).T


def _source_path(src, electric, length):
    """The source as the reference sees it after its first conversion step (fields.py:544-553): a point dipole
    becomes a finite dipole (electric) or a square loop of electric dipoles (magnetic); other formats are unchanged."""
    src = np.asarray(src, dtype=np.float64)
    if src.shape == (5,):
        return _dipole_from_point(src, length) if electric else _loop_from_point(src, length)
    return src


def _source_sign(src, electric):
    """-1 for magnetic sources given as a point dipole or as a path (the reference negates the summed field of an
    arbitrarily shaped source when ``electric=False``, fields.py:574-576); finite dipoles ignore ``electric``."""
    src = np.asarray(src, dtype=np.float64)
    shaped = src.ndim == 2 and src.shape[0] == 3
    return -1.0 if (not electric and (shaped or src.shape == (5,))) else 1.0


def _source_segments(src, strength, length, electric=True):
    """Finite-dipole segments ``[(src6, moment3), ...]`` of a source in any of the reference's formats (point dipole,
    finite dipole, arbitrarily shaped, magnetic point dipole = square loop: emg3d/fields.py:538-600)."""
    if not np.allclose(np.size(src[0]), [np.size(c) for c in src]):
        raise ValueError(f"All source coordinates must have the same dimension.Provided source: {src}.")
    src = _source_path(src, electric, length)
    if src.ndim == 2 and src.shape[0] == 3:            # arbitrarily shaped: one segment per pair of points
        lengths = np.sqrt(np.sum((src[:, :-1] - src[:, 1:]) ** 2, axis=0))
        lengths = lengths / lengths.sum() if strength == 0 else lengths * strength
        out = []
        for i in range(src.shape[1] - 1):
            seg = np.array([src[0, i], src[0, i + 1], src[1, i], src[1, i + 1], src[2, i], src[2, i + 1]])
            out.extend(_source_segments(seg, lengths[i], length))
        return out
    if src.shape != (6,):
        raise ValueError("Source is wrong defined. It must be either\n- a point, [x, y, z, azimuth, dip],\n"
                         "- a finite dipole, [x1, x2, y1, y2, z1, z2], or\n- an arbitrarily shaped dipole, "
                         f"[[x-coo], [y-coo], [z-coo]].\nProvided source: {src}.")
    d = src[1::2] - src[0::2]
    if np.allclose(d, 0, atol=1e-15):
        raise ValueError("Provided finite dipole has no length; use the format [x, y, z, azimuth, dip] instead.")
    moment = d / np.linalg.norm(d) if strength == 0 else strength * d
    return [(src, moment)]


def _spread_dipole(grid, src, comp, decimals):
    """Component ``comp`` of a unit finite dipole on the edges: the reference's ``_finite_source_xyz`` (emg3d/fields.py:
    914-1010) restated -- every cell of the dipole's index bounding box takes the part of the dipole between the parametric
    bounds of that cell (clipped to [0, 1]), located at the midpoint of that part and split between the cell's four
    ``comp``-edges by the bilinear weights there; a cell contributes when the midpoint's weights are non-negative.  For a
    long oblique dipole this includes cells the dipole does not cross, whose contributions the final unity normalisation
    (with the reference's warning) compensates: reproduced as is, the device kernel ``k_source_dipole`` does the same."""
    nodes = [np.round(n, decimals) for n in (grid.nodes_x, grid.nodes_y, grid.nodes_z)]
    src = np.round(np.asarray(src, dtype=np.float64), decimals)
    if any(src[2 * a] < nodes[a][0] or src[2 * a + 1] > nodes[a][-1] for a in range(3)):
        raise ValueError(f"Provided source outside grid: {src}.")
    p0 = src[0::2]
    d = src[1::2] - p0
    inv = d.copy()
    inv[inv != 0] = 1 / inv[inv != 0]
    par = [(nodes[a] - p0[a]) * inv[a] for a in range(3)]          # parametric position of the node planes

    def cell_range(a):
        lo_hi = []
        for v in (min(src[2 * a:2 * a + 2]), max(src[2 * a:2 * a + 2])):
            lo_hi.append(max(0, int(np.where(v < np.r_[nodes[a], np.inf])[0][0]) - 1))
        return lo_hi[0], min(lo_hi[1] + 1, nodes[a].size - 1)

    rng_ = [cell_range(a) for a in range(3)]
    shape = (grid.vnEx, grid.vnEy, grid.vnEz)[comp]
    out = np.zeros(shape, order='F')
    along = d != 0
    slen = np.linalg.norm(d)
    t1, t2 = [a for a in range(3) if a != comp]
    for iz in range(*rng_[2]):
        for iy in range(*rng_[1]):
            for ix in range(*rng_[0]):
                idx = (ix, iy, iz)
                bounds = np.sort(np.vstack([[par[a][idx[a]], par[a][idx[a] + 1]] for a in range(3)])[along, :], 1)
                al = max(0, bounds[:, 0].max())
                ar = min(1, bounds[:, 1].min())
                xmin = p0 + al * d
                xmax = p0 + ar * d
                mid = (xmin + xmax) / 2.0
                part = np.linalg.norm(xmax - xmin) / slen
                r = [(mid[a] - nodes[a][idx[a]]) / grid.h[a][idx[a]] for a in range(3)]
                if min(r) >= 0 and np.max(np.abs(ar - al)) > 0:
                    for b2 in (0, 1):               # the reference's order of the four updates: t1 inner, t2 outer
                        for b1 in (0, 1):
                            ii = list(idx)
                            ii[t1] += b1
                            ii[t2] += b2
                            out[tuple(ii)] += (r[t1] if b1 else 1 - r[t1]) * (r[t2] if b2 else 1 - r[t2]) * part
    total = abs(out.sum())
    if abs(total - 1) > 1e-6:
        msg = f"Normalizing Source: {total:.10f}."
        print(f"* WARNING :: {msg}")
        warnings.warn(msg, UserWarning)
        out /= total
    return out


def get_source_field(grid, src, freq, strength=0, electric=True, length=1.0, decimals=6):
    """Source field ``s mu_0 J_s`` (reference ``fields.get_source_field``, emg3d/fields.py:446-631) built on the host:
    point dipole ``[x, y, z, azimuth, dip]`` (electric: a finite dipole of ``length``; ``electric=False``: a square loop
    of side ``length`` perpendicular to it), finite dipole ``[x0, x1, y0, y1, z0, z1]`` or arbitrarily shaped
    ``[[x-coo], [y-coo], [z-coo]]``; normalised to 1 A m if ``strength=0``.  The device twin is ``DeviceMG.set_source``."""
    strength = np.asarray(strength)
    segs = _source_segments(src, strength, length, electric)
    sign = _source_sign(src, electric)
    path = _source_path(src, electric, length)
    shaped = path.ndim == 2
    sfield = SourceField(grid, freq=freq)
    total = None
    for src6, moment in segs:
        seg = SourceField(grid, freq=freq) if shaped else sfield
        views = (seg.fx, seg.fy, seg.fz)
        d = src6[1::2] - src6[0::2]
        for comp in range(3):
            if d[comp] == 0:
                continue
            views[comp][...] = _spread_dipole(grid, src6, comp, decimals) * (moment[comp] * sfield.smu0)
        if shaped:
            sfield += seg
        total = moment if total is None else total + moment
    if sign < 0:
        sfield *= -1
    sfield.src = path
    sfield.strength = strength
    sfield.moment = total
    return sfield


def get_h_field(grid, model, field):
    """Magnetic field of an electric field by Faraday's law, ``H = -curl E / (s mu_0)`` -- the interface of
    the reference's ``fields.get_h_field`` (emg3d/fields.py:819-911), evaluated by the HIP kernel
    ``k_hfield`` through the C ABI (``emg3d_get_h_field``).

    The returned ``Field`` lives on the faces: ``fx (nNx, nCy, nCz)``, ``fy (nCx, nNy, nCz)``,
    ``fz (nCx, nCy, nNz)``; with ``model.mu_r`` the curl is scaled by the dual-grid average of
    ``zeta = V / mu_r`` over the dual-cell volume (fields.py:878-906)."""
    from . import _lib
    lib = _lib.load()
    smu0 = field.smu0
    if smu0 is None:
        raise ValueError("get_h_field requires a field with a frequency.")
    dtype = np.dtype(np.complex128 if np.iscomplexobj(field) else np.float64)
    if dtype == np.float64 and np.iscomplexobj(smu0):
        dtype = np.dtype(np.complex128)
    e = np.ascontiguousarray(np.asarray(field), dtype=dtype)
    nx, ny, nz = (int(n) for n in grid.vnC)
    if e.size != grid.nE:
        raise ValueError(f"`field` must have grid.nE = {grid.nE} entries; provided: {e.size}.")
    hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
    zeta = None
    if model.mu_r is not None:
        vol = grid.cell_volumes.reshape(grid.vnC, order='F')
        zeta = np.ascontiguousarray((vol / model.mu_r).ravel(order='F'), dtype=np.float64)
    shapes = ((nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1))
    out = np.empty(sum(int(np.prod(sh)) for sh in shapes), dtype=dtype)
    a = complex(smu0)
    _lib.check(lib.emg3d_get_h_field(_lib.dtype_code(dtype), nx, ny, nz, _lib.ptr(out), _lib.ptr(e),
                                     None if zeta is None else _lib.ptr(zeta), _lib.ptr(hx), _lib.ptr(hy),
                                     _lib.ptr(hz), a.real, a.imag), "emg3d_get_h_field")
    return _h_from_vector(out, shapes)


def _h_from_vector(vec, shapes):
    """Wrap a flat ``[hx, hy, hz]`` vector as a magnetic ``Field`` (no frequency, as the reference returns it)."""
    obj = np.asarray(vec).view(Field)
    obj.vnEx, obj.vnEy, obj.vnEz = shapes
    obj.nEx, obj.nEy, obj.nEz = (int(np.prod(sh)) for sh in shapes)
    obj._freq = None
    return obj


def _rotation(azm, dip):
    """Rotation factors (x, y, z) of a dipole: azimuth anti-clockwise from x, dip upwards from the xy-plane, in
    degrees (reference emg3d/fields.py:1013-1034, same SciPy ``cosdg`` / ``sindg``)."""
    from scipy.special import cosdg, sindg
    return np.array([cosdg(azm) * cosdg(dip), sindg(azm) * cosdg(dip), sindg(dip)])


def _receiver_args(rec):
    if len(rec) != 5:
        raise ValueError("`rec` needs to be in the form (x, y, z, azimuth, dip).\n"
                         f"Length of provided `rec`: {len(rec)}.")
    n = max(np.atleast_1d(x).size for x in rec)
    xyz = np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), (n,)) for c in rec[:3]]))
    fac = _rotation(*rec[3:])
    fac = np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(f, dtype=np.float64), (n,)) for f in fac]))
    return n, xyz, fac


class EMArray(np.ndarray):
    """ndarray with ``amp()`` and ``pha()`` (what the reference's ``utils.EMArray`` offers, emg3d/utils.py:117-190)."""

    def __new__(cls, data):
        return np.asarray(data).view(cls)

    def amp(self):
        return np.abs(self.view())

    def pha(self, deg=False, unwrap=True, lag=True):
        pha = np.angle(self.view()) if lag else np.angle(np.conj(self.view()))
        if unwrap and self.size > 1:
            pha = np.unwrap(pha).view(type(self))
        if deg:
            pha = pha * 180 / np.pi
        return pha


def get_receiver(grid, values, coordinates, method='cubic', extrapolate=False):
    """Values of a field component, a whole field (-> tuple ``(fx, fy, fz)``) or a model parameter at ``coordinates =
    (x, y, z)`` (reference ``fields.get_receiver``, emg3d/fields.py:634-730): ``maps.interp3d`` on the grid without its
    first and last point per direction, NaN outside unless ``extrapolate`` (linear: extrapolated; cubic:
    ``map_coordinates``' mode 'nearest').  The interpolation runs on the device."""
    from emg3d_amd import maps
    if hasattr(values, 'field') and values.field.ndim == 1:
        return tuple(get_receiver(grid, f, coordinates, method, extrapolate) for f in (values.fx, values.fy, values.fz))
    if len(coordinates) != 3:
        raise ValueError("Coordinates needs to be in the form (x, y, z).\n"
                         f"Length of provided coord.: {len(coordinates)}.")
    values = np.asarray(values)
    nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)
    centers = (grid.cell_centers_x, grid.cell_centers_y, grid.cell_centers_z)
    points = tuple((nodes[i] if values.shape[i] == grid.vnC[i] + 1 else centers[i])[1:-1] for i in range(3))
    if extrapolate:
        fill_value, mode = None, 'nearest'
    else:
        fill_value, mode = np.array(0, values.dtype) * np.nan, 'constant'
    out = maps.interp3d(points, values[1:-1, 1:-1, 1:-1], coordinates, method, fill_value, mode, cval=np.nan)
    return out if values.size == grid.nC else EMArray(out)


def get_receiver_response(grid, field, rec):
    """Field (response) at point receivers ``rec = (x, y, z, azimuth, dip)`` -- the interface of the reference's
    ``fields.get_receiver_response`` (emg3d/fields.py:733-817): cubic-spline interpolation of every component
    on its trimmed staggered grid (first and last value per direction dropped), NaN outside, components
    weighted with the rotation factors.  Evaluated on the device (``emg3d_get_receiver_response``); for a
    field that already lives in HBM use ``DeviceMG.get_receiver_response`` (no field transfer)."""
    from . import _lib
    if np.ndim(np.asarray(field)) == 3:
        raise ValueError("`field` must be a `Field`-instance, not a\n"
                         "particular field such as `field.fx`.")
    n, xyz, fac = _receiver_args(rec)
    lib = _lib.load()
    dtype = np.dtype(np.complex128 if np.iscomplexobj(field) else np.float64)
    f = np.ascontiguousarray(np.asarray(field), dtype=dtype)
    nx, ny, nz = (int(v) for v in grid.vnC)
    electric = bool(getattr(field, 'is_electric', f.size == grid.nE))
    hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
    origin = np.ascontiguousarray(grid.origin, dtype=np.float64)
    out = np.empty(n, dtype=dtype)
    _lib.check(lib.emg3d_get_receiver_response(_lib.dtype_code(dtype), nx, ny, nz, _lib.ptr(hx), _lib.ptr(hy),
                                               _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(f), int(electric), n,
                                               _lib.ptr(xyz), _lib.ptr(fac), _lib.ptr(out)),
               "emg3d_get_receiver_response")
    return out
