"""Host-side mirror of ``emg3d.solver`` for the multigrid hot path.

Same entry points, argument meaning, return values and error behaviour as the
reference (emg3d/solver.py): :func:`solve`, :func:`multigrid`, :func:`krylov`,
:func:`smoothing`, :func:`restriction`, :func:`prolongation`,
:func:`residual`, :class:`MGParameters` and the small helpers.  The numerics
run on the GPU through the C ABI (``include/emg3d_hip.h``):

* ``solve`` / ``multigrid`` / ``krylov`` keep grids, model, fields and the
  cached line factorisations device-resident in a :class:`DeviceMG` handle
  (tier 2); the host only sees one residual norm per cycle.
* the stand-alone sub-routines operate on host arrays (tier 1), like the
  reference's wrappers around ``emg3d.core``.

Build-specific keyword: ``ordering`` = ``'colour'`` (default; 4-/8-colour
Gauss-Seidel, the throughput mode) or ``'lex'`` (the reference's lexicographic
update order executed as hyperplane wavefronts; cycle-by-cycle parity).
"""
import ctypes
import itertools
import time
from dataclasses import dataclass
from datetime import datetime, timedelta

import warnings

import numpy as np
import scipy.sparse.linalg as ssl

from emg3d_amd import _lib, core, fields, meshes, models

__all__ = ['solve', 'multigrid', 'krylov', 'smoothing', 'restriction', 'prolongation',
           'residual', 'MGParameters', 'DeviceMG']

ORDERINGS = {'lex': 0, 'colour': 1, 'color': 1}


# --------------------------------------------------------------------------
# Device handle
# --------------------------------------------------------------------------
class DeviceMG:
    """Device-resident multigrid state (C-ABI tier 2, ``emg3d_mg_*``)."""

    def __init__(self, grid, vmodel, dtype, device=0):
        self._lib = _lib.load()
        self.dtype = np.dtype(dtype)
        self.nE = int(grid.nE)
        self.nC = int(grid.nC)
        self.device = int(device)
        code = _lib.dtype_code(self.dtype)
        hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
        origin = np.ascontiguousarray(grid.origin, dtype=np.float64)

        def cells(a, dt):
            return np.ascontiguousarray(np.asarray(a, dtype=dt).ravel(order='F'))
        etx = cells(vmodel.eta_x, self.dtype)
        ety = etx if vmodel.eta_y is vmodel.eta_x else cells(vmodel.eta_y, self.dtype)
        etz = etx if vmodel.eta_z is vmodel.eta_x else cells(vmodel.eta_z, self.dtype)
        zeta = cells(vmodel.zeta, np.float64)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.emg3d_mg_create(
            ctypes.byref(handle), code, *(int(n) for n in grid.vnC), _lib.ptr(hx), _lib.ptr(hy),
            _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(etx), _lib.ptr(ety), _lib.ptr(etz),
            _lib.ptr(zeta), int(device)), "emg3d_mg_create")
        self._h = handle

    @classmethod
    def from_sigma_volume(cls, grid, sv_x, sv_y, sv_z, zeta, smu0, device=0):
        """Handle from the frequency-independent model (``models.sigma_volume``) and the scalar
        ``smu0 = s*mu_0``: ``eta = smu0 * sv`` is formed on the device (``emg3d_mg_create_sv``)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.dtype = np.dtype(np.complex128 if np.iscomplexobj(smu0) else np.float64)
        self.nE = int(grid.nE)
        self.nC = int(grid.nC)
        self.device = int(device)
        hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
        origin = np.ascontiguousarray(grid.origin, dtype=np.float64)

        def cells(a):
            return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order='F'))
        svx = cells(sv_x)
        svy = svx if sv_y is sv_x else cells(sv_y)
        svz = svx if sv_z is sv_x else cells(sv_z)
        zt = cells(zeta)
        handle = ctypes.c_void_p()
        a = complex(smu0)
        _lib.check(self._lib.emg3d_mg_create_sv(
            ctypes.byref(handle), _lib.dtype_code(self.dtype), *(int(n) for n in grid.vnC), _lib.ptr(hx),
            _lib.ptr(hy), _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(svx), _lib.ptr(svy), _lib.ptr(svz),
            _lib.ptr(zt), a.real, a.imag, int(device)), "emg3d_mg_create_sv")
        self._h = handle
        return self

    @classmethod
    def from_model_parts(cls, grid, sigma_x, sigma_y, sigma_z, vol, zeta, resistivity=False, smu0=None, device=0,
                         epsilon_r=None, sval=None):
        """Handle from ``models.model_parts`` and ``smu0 = s*mu_0``: ``eta = (smu0 * vol) * sigma`` is formed on the device
        with VolumeModel's rounding (``emg3d_mg_create_vs``) -- bit for bit the reference's eta at this and, after
        ``set_smu0``, at every other frequency.  ``resistivity=True``: the three arrays hold resistivities
        (``models.model_parts(..., raw=True)``), the device takes the reciprocal.  ``epsilon_r`` (``parts.epsilon_r``) with
        ``sval = s``: ``eta = (smu0 vol) (sigma - s eps_0 eps_r)`` (``emg3d_mg_create_vse``, reference models.py:639-647)."""
        if smu0 is None:
            raise TypeError("from_model_parts: smu0 is required.")
        if epsilon_r is not None and sval is None:
            raise TypeError("from_model_parts: epsilon_r needs sval (s = 2 i pi f resp. the Laplace parameter).")
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.dtype = np.dtype(np.complex128 if np.iscomplexobj(smu0) else np.float64)
        self.nE = int(grid.nE)
        self.nC = int(grid.nC)
        self.device = int(device)
        hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
        origin = np.ascontiguousarray(grid.origin, dtype=np.float64)

        def cells(a):
            return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order='F'))
        sx = cells(sigma_x)
        sy = sx if sigma_y is sigma_x else cells(sigma_y)
        sz = sx if sigma_z is sigma_x else cells(sigma_z)
        vl, zt = cells(vol), cells(zeta)
        handle = ctypes.c_void_p()
        a = complex(smu0)
        self._eps = epsilon_r is not None
        if self._eps:
            ep = cells(epsilon_r)
            _lib.check(self._lib.emg3d_mg_create_vse(
                ctypes.byref(handle), _lib.dtype_code(self.dtype), *(int(n) for n in grid.vnC), _lib.ptr(hx),
                _lib.ptr(hy), _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(sx), _lib.ptr(sy), _lib.ptr(sz), _lib.ptr(vl),
                _lib.ptr(zt), _lib.ptr(ep), a.real, a.imag, models.seps0_of(sval), int(bool(resistivity)), int(device)),
                "emg3d_mg_create_vse")
        else:
            _lib.check(self._lib.emg3d_mg_create_vs(
                ctypes.byref(handle), _lib.dtype_code(self.dtype), *(int(n) for n in grid.vnC), _lib.ptr(hx),
                _lib.ptr(hy), _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(sx), _lib.ptr(sy), _lib.ptr(sz), _lib.ptr(vl),
                _lib.ptr(zt), a.real, a.imag, int(bool(resistivity)), int(device)), "emg3d_mg_create_vs")
        self._h = handle
        return self

    @classmethod
    def from_model(cls, grid, parts, spec, device=0):
        """``from_model_parts`` for ``parts = models.model_parts(grid, model, raw=True)`` and a field / frequency object
        ``spec`` (``smu0``, ``sval``): with or without ``epsilon_r``."""
        return cls.from_model_parts(grid, *parts, smu0=spec.smu0, device=device, epsilon_r=getattr(parts, 'epsilon_r', None),
                                    sval=spec.sval)

    def set_smu0(self, smu0, sval=None):
        """Re-target a ``from_sigma_volume`` handle to another frequency (``emg3d_mg_set_smu0``): eta, the coarse models,
        every cached line factorisation are recomputed on the device as a fresh handle would; grids, work buffers and
        launch graphs stay.  Same dtype only (a Laplace-domain handle takes a real ``smu0``).  Handles with ``epsilon_r``
        need ``sval`` too (``emg3d_mg_set_smu0_eps``)."""
        a = complex(smu0)
        if self.dtype == np.float64 and a.imag != 0.0:
            raise ValueError("set_smu0: a float64 (Laplace-domain) handle takes a real s*mu_0.")
        if self.dtype == np.complex128 and not np.iscomplexobj(smu0):
            raise ValueError("set_smu0: a complex128 (frequency-domain) handle takes a complex s*mu_0.")
        if getattr(self, '_eps', False):
            if sval is None:
                raise TypeError("set_smu0: a handle with epsilon_r needs sval.")
            _lib.check(self._lib.emg3d_mg_set_smu0_eps(self._h, a.real, a.imag, models.seps0_of(sval)), "emg3d_mg_set_smu0_eps")
            return
        _lib.check(self._lib.emg3d_mg_set_smu0(self._h, a.real, a.imag), "emg3d_mg_set_smu0")

    def get_hfield(self, grid, smu0, mu_r=False):
        """``fields.get_h_field`` (reference fields.py:819-911) of the device-resident electric field:
        the curl runs on the device, only H crosses PCIe.  ``mu_r``: the model has ``mu_r``."""
        nx, ny, nz = (int(n) for n in grid.vnC)
        shapes = ((nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1))
        out = np.empty(sum(int(np.prod(sh)) for sh in shapes), dtype=self.dtype)
        a = complex(smu0)
        _lib.check(self._lib.emg3d_mg_get_hfield(self._h, int(bool(mu_r)), a.real, a.imag, _lib.ptr(out)),
                   "emg3d_mg_get_hfield")
        return fields._h_from_vector(out, shapes)

    def get_receiver_response(self, rec, magnetic=False, smu0=None, mu_r=False):
        """``fields.get_receiver_response`` (reference fields.py:733-817) of the DEVICE-RESIDENT electric field
        (``magnetic=True``: of ``H = get_h_field(E)``, formed on the device; needs ``smu0``): spline
        prefilter and evaluation run on the device, 16 bytes per receiver cross PCIe."""
        n, xyz, fac = fields._receiver_args(rec)
        out = np.empty(n, dtype=self.dtype)
        a = complex(smu0) if smu0 is not None else 0j
        if magnetic and smu0 is None:
            raise ValueError("magnetic receivers need `smu0` (field.smu0).")
        _lib.check(self._lib.emg3d_mg_get_receiver_response(self._h, int(bool(magnetic)), int(bool(mu_r)), a.real,
                                                            a.imag, n, _lib.ptr(xyz), _lib.ptr(fac), _lib.ptr(out)),
                   "emg3d_mg_get_receiver_response")
        return out

    def gradient(self, efield_vec, smu0):
        """Adjoint-state gradient of one (source, frequency) pair on this grid (reference optimize.py:176-199):
        the handle's field = back-propagated field, workspace vector ``efield_vec`` = forward field."""
        out = np.empty(self.nC, dtype=np.float64)
        a = complex(smu0)
        _lib.check(self._lib.emg3d_mg_gradient(self._h, int(efield_vec), a.real, a.imag, _lib.ptr(out)),
                   "emg3d_mg_gradient")
        return out

    def set_source(self, src, smu0, strength=0, length=1.0, decimals=6, accumulate=False, electric=True):
        """Build the source field ``s mu_0 J_s`` of an electric source IN HBM (``fields.get_source_field``,
        reference emg3d/fields.py:446-631): ``src`` = point dipole ``[x, y, z, azimuth, dip]``, finite dipole
        ``[x0, x1, y0, y1, z0, z1]`` or arbitrarily shaped ``[[x-coo], [y-coo], [z-coo]]``; ``electric=False``: a
        magnetic point dipole (a square loop of side ``length``, fields.py:1043-1049).  The edge
        distribution (``_finite_source_xyz``, fields.py:914-1010) runs on the device; six coordinates per
        segment cross PCIe instead of the nE-sized field.  Returns the moment (sum over segments)."""
        segs = fields._source_segments(src, strength, length, electric)
        a = complex(smu0) * fields._source_sign(src, electric)      # magnetic (loop) sources: the field is negated
        total = 0
        for k, (src6, moment) in enumerate(segs):
            sc = np.asarray(moment, dtype=np.complex128) * a
            scale = np.ascontiguousarray(np.stack([sc.real, sc.imag], axis=1).ravel())
            s6 = np.ascontiguousarray(src6, dtype=np.float64)
            sums = np.zeros(3)
            st = self._lib.emg3d_mg_set_sfield_dipole(self._h, _lib.ptr(s6), _lib.ptr(scale), int(decimals),
                                                      int(k > 0 or accumulate), _lib.ptr(sums))
            if st == -4:
                raise ValueError(f"Provided source outside grid: {np.round(s6, decimals)}.")
            _lib.check(st, "emg3d_mg_set_sfield_dipole")
            # "Ensure unity" (fields.py:1003-1010): the device has divided the weights by |sum|; the reference's
            # print + UserWarning are raised here, per component with a moment
            r6 = np.round(s6, decimals)
            for c in range(3):
                sum_s = abs(sums[c])
                if r6[2 * c + 1] != r6[2 * c] and abs(sum_s - 1) > 1e-6:
                    msg = f"Normalizing Source: {sum_s:.10f}."
                    print(f"* WARNING :: {msg}")
                    warnings.warn(msg, UserWarning)
            total = total + moment
        return total

    def set_sfield_vector(self, vector, smu0):
        """s = smu0 * vector with the real source vector (``SourceField.vector``), scaled on the device."""
        v = np.ascontiguousarray(np.asarray(vector), dtype=np.float64)
        a = complex(smu0)
        _lib.check(self._lib.emg3d_mg_set_sfield_vector(self._h, _lib.ptr(v), a.real, a.imag),
                   "emg3d_mg_set_sfield_vector")

    def close(self):
        if getattr(self, '_h', None):
            self._lib.emg3d_mg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_params(self, var, ordering=None):
        clevel = (ctypes.c_int * 4)(*[int(c) for c in var.clevel])
        order = ORDERINGS[var.ordering if ordering is None else ordering]
        cyc = ord(var.cycle) if var.cycle else ord('V')
        _lib.check(self._lib.emg3d_mg_set_params(self._h, cyc, int(var.nu_init), int(var.nu_pre),
                                                 int(var.nu_coarse), int(var.nu_post), clevel, order),
                   "emg3d_mg_set_params")

    def set_trace(self, on):
        """verb = 5: follow every smoothing call of every level with a residual norm (cycles then launch eagerly)."""
        _lib.check(self._lib.emg3d_mg_set_trace(self._h, int(bool(on))), "emg3d_mg_set_trace")

    def get_trace(self, max_recs=16384):
        """Records collected since the last call: [(it, level, cycmax, kind, (nx, ny, nz), norm)], kind 0 = coarsest level,
        1 = pre-, 2 = post-smoothing; it = -1 on level 0."""
        recs = np.zeros((max_recs, 7), dtype=np.int64)
        norms = np.zeros(max_recs, dtype=np.float64)
        n = ctypes.c_int(0)
        _lib.check(self._lib.emg3d_mg_get_trace(self._h, int(max_recs), _lib.ptr(recs), _lib.ptr(norms), ctypes.byref(n)),
                   "emg3d_mg_get_trace")
        return [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), (int(r[4]), int(r[5]), int(r[6])), float(v))
                for r, v in zip(recs[:n.value], norms[:n.value])]

    def prepare(self, sc_dir, lr_dir):
        """Build everything loop invariant for cycles with this (sc_dir, lr_dir) without running one."""
        _lib.check(self._lib.emg3d_mg_prepare(self._h, int(sc_dir), int(lr_dir)), "emg3d_mg_prepare")

    def _field(self, f):
        return np.ascontiguousarray(np.asarray(f), dtype=self.dtype)

    def set_sfield(self, sfield):
        s = self._field(sfield)
        _lib.check(self._lib.emg3d_mg_set_sfield(self._h, _lib.ptr(s)), "emg3d_mg_set_sfield")

    def set_efield(self, efield=None):
        if efield is None:
            _lib.check(self._lib.emg3d_mg_set_efield(self._h, None), "emg3d_mg_set_efield")
        else:
            e = self._field(efield)
            _lib.check(self._lib.emg3d_mg_set_efield(self._h, _lib.ptr(e)), "emg3d_mg_set_efield")

    def get_efield(self, out=None):
        if out is None:
            out = np.empty(self.nE, dtype=self.dtype)
        buf = out if (isinstance(out, np.ndarray) and out.flags.c_contiguous and out.dtype == self.dtype) \
            else np.empty(self.nE, dtype=self.dtype)
        _lib.check(self._lib.emg3d_mg_get_efield(self._h, _lib.ptr(buf)), "emg3d_mg_get_efield")
        if buf is not out:
            out[...] = buf
        return out

    def get_residual(self):
        out = np.empty(self.nE, dtype=self.dtype)
        _lib.check(self._lib.emg3d_mg_get_residual(self._h, _lib.ptr(out)), "emg3d_mg_get_residual")
        return out

    # ---- batched systems: several sources through the same launches (include/emg3d_hip.h) ----
    @property
    def nsys(self):
        return int(self._lib.emg3d_mg_get_batch(self._h))

    def set_batch(self, n):
        """Carry ``n`` systems (right-hand sides on this grid, model and frequency) through every launch of the
        cycle.  Before the first cycle only; zeroes the fields."""
        st = self._lib.emg3d_mg_set_batch(self._h, int(n))
        if st == -6:
            raise RuntimeError("set_batch: the handle has already run or prepared a cycle.")
        _lib.check(st, "emg3d_mg_set_batch")

    def select(self, b):
        """The system that set/get sfield/efield, set_source, get_hfield, get_receiver_response, get_residual,
        sfield_norm and gradient address from now on."""
        _lib.check(self._lib.emg3d_mg_select(self._h, int(b)), "emg3d_mg_select")

    def set_mask(self, active):
        """``active[b] == 0`` freezes system b: its cycles are skipped, its field stays as it is."""
        a = np.ascontiguousarray(active, dtype=np.int32)
        if a.size != self.nsys:
            raise ValueError(f"set_mask: {a.size} flags for {self.nsys} systems.")
        _lib.check(self._lib.emg3d_mg_set_mask(self._h, _lib.ptr(a)), "emg3d_mg_set_mask")

    def _norms(self, fn, name, *args):
        n = self.nsys
        out = np.empty(n, dtype=np.float64)
        _lib.check(fn(self._h, *args, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), name)
        return float(out[0]) if n == 1 else out

    def residual_norm(self):
        """||s - A e||_2 -- a float, or one value per system of a batch."""
        return self._norms(self._lib.emg3d_mg_residual_norm, "emg3d_mg_residual_norm")

    def sfield_norm(self):
        v = ctypes.c_double()
        _lib.check(self._lib.emg3d_mg_sfield_norm(self._h, ctypes.byref(v)), "emg3d_mg_sfield_norm")
        return v.value

    def smooth(self, nu, lr_dir):
        _lib.check(self._lib.emg3d_mg_smooth(self._h, int(nu), int(lr_dir)), "emg3d_mg_smooth")

    def begin(self, sc_dir):
        """Entry of ``solver.multigrid``: level 0's cycmax is fixed from the sc_dir of this moment, as the reference
        does (solver.py:480-485), for all cycles of the call."""
        _lib.check(self._lib.emg3d_mg_begin(self._h, int(sc_dir)), "emg3d_mg_begin")

    def cycle(self, sc_dir, lr_dir, nxt=None):
        """One multigrid cycle; returns the end-of-cycle residual norm (one per system of a batch).  ``nxt`` = the
        (sc_dir, lr_dir) of the NEXT cycle: its loop-invariant set-up then runs on the host while the device works."""
        if nxt is not None:
            return self._norms(self._lib.emg3d_mg_cycle_next, "emg3d_mg_cycle_next", int(sc_dir), int(lr_dir),
                               int(nxt[0]), int(nxt[1]))
        return self._norms(self._lib.emg3d_mg_cycle, "emg3d_mg_cycle", int(sc_dir), int(lr_dir))

    def cycles(self, n, sc_cycle, lr_cycle):
        """``n`` cycles back to back; norms of shape (n,), or (n, nsys) for a batch."""
        sc = np.ascontiguousarray(sc_cycle, dtype=np.int32)
        lr = np.ascontiguousarray(lr_cycle, dtype=np.int32)
        ns = self.nsys
        out = np.empty(n * ns, dtype=np.float64)
        _lib.check(self._lib.emg3d_mg_cycles(self._h, int(n), _lib.ptr(sc), sc.size, _lib.ptr(lr),
                                             lr.size, _lib.ptr(out)), "emg3d_mg_cycles")
        return out if ns == 1 else out.reshape(n, ns)

    def amatvec(self, x):
        x = self._field(x)
        y = np.empty(self.nE, dtype=self.dtype)
        _lib.check(self._lib.emg3d_mg_amatvec(self._h, _lib.ptr(x), _lib.ptr(y)), "emg3d_mg_amatvec")
        return y

    # ---- device-resident vector workspace (Krylov iteration) ----
    SFIELD, EFIELD = -1, -2      # vector ids of the level-0 source / field

    def vec_alloc(self, n):
        _lib.check(self._lib.emg3d_mg_vec_alloc(self._h, int(n)), "emg3d_mg_vec_alloc")

    def vec_set(self, i, x):
        x = self._field(x)
        _lib.check(self._lib.emg3d_mg_vec_set(self._h, int(i), _lib.ptr(x)), "emg3d_mg_vec_set")

    def vec_get(self, i):
        y = np.empty(self.nE, dtype=self.dtype)
        _lib.check(self._lib.emg3d_mg_vec_get(self._h, int(i), _lib.ptr(y)), "emg3d_mg_vec_get")
        return y

    def vec_copy(self, dst, src):
        _lib.check(self._lib.emg3d_mg_vec_copy(self._h, int(dst), int(src)), "emg3d_mg_vec_copy")

    def vec_axpy(self, y, alpha, x):
        """y += alpha * x"""
        a = complex(alpha)
        _lib.check(self._lib.emg3d_mg_vec_axpy(self._h, int(y), a.real, a.imag, int(x)), "emg3d_mg_vec_axpy")

    def vec_scale(self, y, alpha):
        a = complex(alpha)
        _lib.check(self._lib.emg3d_mg_vec_scale(self._h, int(y), a.real, a.imag), "emg3d_mg_vec_scale")

    def vec_dot(self, a, b):
        """numpy.vdot(a, b) (first argument conjugated) / numpy.dot for float64."""
        out = (ctypes.c_double * 2)()
        _lib.check(self._lib.emg3d_mg_vec_dot(self._h, int(a), int(b), out), "emg3d_mg_vec_dot")
        return complex(out[0], out[1]) if self.dtype.kind == 'c' else float(out[0])

    def vec_norm(self, a):
        return float(np.sqrt(abs(self.vec_dot(a, a))))

    def vec_amatvec(self, dst, src):
        _lib.check(self._lib.emg3d_mg_vec_amatvec(self._h, int(dst), int(src)), "emg3d_mg_vec_amatvec")

    def time_sweep(self, direction, reps=3):
        v = ctypes.c_float()
        _lib.check(self._lib.emg3d_mg_time_sweep(self._h, int(direction), int(reps), ctypes.byref(v)),
                   "emg3d_mg_time_sweep")
        return v.value

    def placement(self):
        """Record of the placement of level 0's working copies (``emg3d_mg_placement``): per working copy ('x': the transposed
        copy the x-line sweeps write, 'yz': the x-split copy of the y- / z-line sweeps) the candidates timed, the index of the one
        kept (0 = the block the handle had) and ms per sweep of every candidate; {} when nothing was placed (small levels,
        ``EMG3D_PLACE_TRIES=0``)."""
        out = {}
        for w, name in ((0, 'x'), (1, 'yz')):
            tries, kept = ctypes.c_int(0), ctypes.c_int(0)
            ms = (ctypes.c_float * 16)()
            _lib.check(self._lib.emg3d_mg_placement(self._h, w, ctypes.byref(tries), ctypes.byref(kept), ms), "emg3d_mg_placement")
            if kept.value < 0:
                out[name] = {"tries": 0, "kept": -1, "reused": True}        # (the block an earlier handle of the process had found)
            elif tries.value > 0:
                out[name] = {"tries": tries.value, "kept": kept.value, "first_ms": float(ms[0]), "kept_ms": float(ms[kept.value]),
                             "ms_per_sweep": [float(ms[k]) for k in range(tries.value)]}
        return out

    def last_sweep_kernel(self):
        """Name of the kernel instantiation the most recent line-sweep launch of this handle selected."""
        buf = ctypes.create_string_buffer(64)
        _lib.check(self._lib.emg3d_mg_last_sweep_kernel(self._h, buf), "emg3d_mg_last_sweep_kernel")
        return buf.value.decode()

    def last_residual_kernel(self):
        """Name of the kernel instantiation the most recent residual launch of this handle selected."""
        buf = ctypes.create_string_buffer(64)
        _lib.check(self._lib.emg3d_mg_last_residual_kernel(self._h, buf), "emg3d_mg_last_residual_kernel")
        return buf.value.decode()

    def time_residual(self, reps=3):
        v = ctypes.c_float()
        _lib.check(self._lib.emg3d_mg_time_residual(self._h, int(reps), ctypes.byref(v)),
                   "emg3d_mg_time_residual")
        return v.value

    @property
    def device_bytes(self):
        return int(self._lib.emg3d_mg_device_bytes(self._h))

    @property
    def efield_devptr(self):
        return self._lib.emg3d_mg_efield_devptr(self._h)

    @property
    def stream_ptr(self):
        """The handle's HIP stream (``hipStream_t`` as an integer) for ``torch.cuda.ExternalStream``."""
        return int(self._lib.emg3d_mg_stream(self._h) or 0)


# --------------------------------------------------------------------------
# solve
# --------------------------------------------------------------------------
def solve(grid, model, sfield, efield=None, cycle='F', sslsolver=False, semicoarsening=False,
          linerelaxation=False, verb=1, **kwargs):
    """Solve Maxwell's equations with multigrid and/or a Krylov solver.

    Mirrors ``emg3d.solver.solve`` (reference emg3d/solver.py:35-430): same
    parameters (``tol, maxit, nu_init, nu_pre, nu_coarse, nu_post, clevel,
    return_info, log``), same return convention (``efield`` if none was
    provided, ``info_dict`` if ``return_info``), same ``info_dict`` keys and
    exit messages.  Extra keywords: ``ordering`` ('colour'|'lex'), ``device``,
    ``handle`` (an existing ``DeviceMG`` for this grid/model/frequency, e.g. from
    ``DeviceMG.from_sigma_volume``; it is used as is and not closed; ``model`` may then be None),
    ``source=(src, strength)`` (the source is built in HBM by ``DeviceMG.set_source`` -- ``sfield`` then only
    carries the frequency and is not uploaded; ``source='resident'``: the handle already holds it),
    ``download=False`` (multigrid only: the solution stays in HBM, the returned field is None).
    """
    device = kwargs.pop('device', 0)
    handle = kwargs.pop('handle', None)
    source = kwargs.pop('source', None)     # (src, strength): build the source in HBM instead of uploading `sfield`;
    #                                         'resident': the handle already holds the source
    download = kwargs.pop('download', True)  # False: leave the solution in HBM (returned efield is None)
    var = MGParameters(cycle=cycle, sslsolver=sslsolver, semicoarsening=semicoarsening,
                       linerelaxation=linerelaxation, vnC=grid.vnC, verb=verb, **kwargs)

    var.cprint(f"\n:: emg3d START :: {var.time.now} :: emg3d_amd (MI355X/HIP)\n", 2)
    var.cprint(var, 2)

    if sfield.freq is None:
        raise ValueError("Source field is missing frequency information;\n"
                         "Create it with `emg3d_amd.fields.get_source_field`, or\n"
                         "initiate it with `emg3d_amd.fields.SourceField`.")

    info = ""
    if handle is None:
        parts = _exact_parts(grid, model, sfield.smu0)
        if parts is not None:
            # eta = (smu0 V) sigma bit for bit as VolumeModel would give it, formed on the device from sigma and V
            vmodel = None
            dev = DeviceMG.from_model(grid, parts, sfield, device=device)
        else:
            vmodel = models.VolumeModel(grid, model, sfield)
            dev = DeviceMG(grid, vmodel, sfield.dtype, device=device)
    else:
        vmodel = None       # the device handle holds eta, zeta
        dev = handle
        if dev.dtype != sfield.dtype:
            raise ValueError(f"`handle` is {dev.dtype}, the source field {sfield.dtype}.")
    try:
        dev.set_params(var)
        if isinstance(source, str) and source == 'resident':
            if var.sslsolver:
                sfield.field[:] = dev.vec_get(dev.SFIELD)
        elif source is not None:
            # the source field never exists on the host: six coordinates per dipole segment go up, the edge
            # distribution runs on the device (DeviceMG.set_source); `sfield` only carries the frequency
            dev.set_source(source[0], sfield.smu0, strength=source[1] if len(source) > 1 else 0,
                           electric=source[2] if len(source) > 2 else True)
            if var.sslsolver:               # the Krylov drivers take the right-hand side from the host object
                sfield.field[:] = dev.vec_get(dev.SFIELD)
        else:
            dev.set_sfield(sfield)
        # ||sfield||_2 (reference solver.py:305, scipy.linalg.norm) on the device, from the copy just uploaded.
        # Not numpy/BLAS on the host: the 64-128 worker threads a multi-threaded BLAS spins up for this one
        # norm stall the GPU queues of the process once, 30-50 ms later, for 60-80 ms (tools/idle_gap.py:
        # 128^3 solve 0.20 s with the host norm, 0.14 s without).
        var.l2_refe = dev.sfield_norm()
        var.error_at_cycle[0] = var.l2_refe

        if efield is None:
            efield = fields.Field(grid, dtype=sfield.dtype, freq=sfield._freq)
            var.do_return = True
            dev.set_efield(None)
        else:
            if sfield.dtype != efield.dtype:
                raise ValueError("Source field and electric field must have the\n same dtype; "
                                 "complex (f-domain) or real (s-domain).\n Provided:"
                                 f"sfield: {sfield.dtype}; efield: {efield.dtype}.")
            if efield.freq is None:
                efield._freq = sfield._freq
            var.do_return = False
            dev.set_efield(efield)
            var.l2 = dev.residual_norm()
            if var.l2 < var.tol * var.l2_refe:
                var.sslsolver = None
                var.cycle = None
                var.exit_message = "CONVERGED"
                info = "   > NOTHING DONE (provided efield already good enough)\n"

        if var.l2_refe < 100 * np.finfo(float).tiny:
            var.l2_refe = np.nan
            var.sslsolver = None
            var.cycle = None
            var.exit_message = "CONVERGED"
            info = "   > RETURN ZERO E-FIELD (provided sfield is zero)\n"
            efield = fields.Field(grid, dtype=sfield.dtype, freq=sfield._freq)

        header = f"   [hh:mm:ss]  {'rel. error':<22}"
        if var.sslsolver:
            header += f"{'solver':<20}"
            if var.cycle:
                header += f"{'MG':<11} l s"
            var.cprint(header + "\n", 3)
        elif var.cycle:
            var.cprint(header + f"{'[abs. error, last/prev]':>29}   l s\n", 3)

        if var.sslsolver:
            krylov(grid, vmodel, sfield, efield, var, dev=dev)
        elif var.cycle:
            var._dev_efield_current = True
            multigrid(grid, vmodel, sfield, efield if download else None, var, dev=dev)
            if not download and var.do_return:
                efield = None
    finally:
        if handle is None:
            dev.close()

    exit_status = int(var.exit_message != 'CONVERGED')

    if var.verb < 0 or var.verb == 2:
        var.one_liner(var.l2, True)
    elif var.verb > 2:
        if var.sslsolver:
            info = f"   > Solver steps     : {var._ssl_it}\n"
            if var.cycle:
                info += f"   > MG prec. steps   : {var.it}\n"
        elif var.cycle:
            info = f"   > MG cycles        : {var.it}\n"
        info += f"   > Final rel. error : {var.l2/var.l2_refe:.3e}\n\n"
        info += f":: emg3d END   :: {var.time.now} :: runtime = {var.time.runtime}\n"
        var.cprint(info, 2)
    elif var.verb == 1 and exit_status == 1:
        var.cprint(f"* WARNING :: {var.exit_message}", 0)

    if var.return_info:
        info_dict = {
            'exit': exit_status,
            'exit_message': var.exit_message,
            'abs_error': var.l2,
            'rel_error': var.l2 / var.l2_refe,
            'ref_error': var.l2_refe,
            'tol': var.tol,
            'it_mg': var.it,
            'it_ssl': var._ssl_it,
            'time': var.runtime_at_cycle[-1],
            'runtime_at_cycle': var.runtime_at_cycle,
            'error_at_cycle': var.error_at_cycle,
            'log': var.log_message,
        }

    if var.do_return and var.return_info:
        return efield, info_dict
    elif var.do_return:
        return efield
    elif var.return_info:
        return info_dict


def _exact_parts(grid, model, smu0):
    """``models.model_parts`` where the device can form VolumeModel's eta from them: s*mu_0 purely imaginary (frequency
    domain) or real (Laplace domain); with or without epsilon_r."""
    if np.iscomplexobj(smu0) and np.real(smu0) != 0.0:
        return None
    return models.model_parts(grid, model, raw=True)


def solve_sources(grid, model, sources, frequency, strength=0, cycle='F', semicoarsening=False,
                  linerelaxation=False, verb=1, rec=None, download=True, electric=True, **kwargs):
    """``[solve(grid, model, get_source_field(grid, src, frequency, strength), ...) for src in sources]`` as ONE
    batched multigrid iteration: the sources of a survey share grid, model and frequency (the reference loops over
    them one solve at a time, simulations.py:916-1015), hence the operator, the coarse models and the cached
    line factorisations; the device carries all of them through every launch of the cycle (``DeviceMG.set_batch``).
    Every system goes through the arithmetic of a solve of its own and stops by its own termination tests
    (a finished system is frozen, ``DeviceMG.set_mask``): fields and ``info_dict`` equal those of separate solves.

    ``sources``: electric sources as ``DeviceMG.set_source`` takes them (built in HBM), or ``SourceField`` objects
    of frequency ``frequency`` (uploaded).  Multigrid only (``sslsolver`` is not batched).  ``rec``: receivers
    ``(x, y, z, azimuth, dip)`` -- the responses are extracted on the device.  ``download=False``: no fields
    returned.  Returns ``(efields | None, info_dicts)`` and the responses ``(n_sources, n_rec)`` if ``rec``.
    """
    if kwargs.get('sslsolver'):
        raise ValueError("solve_sources batches multigrid cycles; use solve() per source with a Krylov solver.")
    kwargs.pop('sslsolver', None)
    device = kwargs.pop('device', 0)
    handle = kwargs.pop('handle', None)       # an existing DeviceMG of this grid / model / frequency: used, not closed
    n = len(sources)
    if n < 1:
        raise ValueError("solve_sources: no sources.")
    proto = fields.FrequencySpec(frequency)                 # dtype, smu0 of this frequency (no nE-sized array)
    host_fields = [s if hasattr(s, 'field') else None for s in sources]
    for sf in host_fields:
        if sf is not None and (sf.freq is None or sf._freq != proto._freq):
            raise ValueError("solve_sources: every source field must carry the frequency of the batch.")
    vars_ = [MGParameters(cycle=cycle, sslsolver=False, semicoarsening=semicoarsening,
                          linerelaxation=linerelaxation, vnC=grid.vnC, verb=verb, **kwargs) for _ in range(n)]
    v0 = vars_[0]
    if handle is not None:
        # the caller's handle, already on this frequency (DeviceMG.set_smu0) and fresh or batched for exactly n systems
        dev = handle
        if dev.dtype != proto.dtype:
            raise ValueError(f"solve_sources: `handle` is {dev.dtype}, the frequency needs {proto.dtype}.")
    else:
        parts = _exact_parts(grid, model, proto.smu0)
        if parts is not None:
            dev = DeviceMG.from_model(grid, parts, proto, device=device)
        else:
            dev = DeviceMG(grid, models.VolumeModel(grid, model, proto), proto.dtype, device=device)
    try:
        dev.set_params(v0)
        if dev.nsys != n:
            dev.set_batch(n)
        active = np.ones(n, dtype=np.int32)
        if handle is not None:
            dev.set_mask(active)
        for b, (src, var) in enumerate(zip(sources, vars_)):
            dev.select(b)
            if handle is not None:
                dev.set_efield(None)
            if host_fields[b] is not None:
                dev.set_sfield(host_fields[b])
            else:
                dev.set_source(src, proto.smu0, strength=strength, electric=electric)
            var.l2_refe = dev.sfield_norm()
            var.error_at_cycle[0] = var.l2_refe
            var.do_return = True
            if var.l2_refe < 100 * np.finfo(float).tiny:        # zero source: zero field (solver.py:330-337)
                var.l2_refe = np.nan
                var.exit_message = "CONVERGED"
                var.l2 = 0.0
                active[b] = 0
        if not active.all():
            dev.set_mask(active)
        dev.begin(v0.sc_dir)
        l2_last = np.atleast_1d(dev.residual_norm()).copy()
        l2_stag = [np.ones(v0._maxcycle) * l2_last[b] for b in range(n)]
        if v0.nu_init > 0 and active.any():
            dev.smooth(v0.nu_init, v0.lr_dir)
        it = 0
        while active.any():
            l2_prev = l2_last.copy()
            for b in range(n):
                l2_stag[b][(it - 1) % v0._maxcycle] = l2_last[b]
            nxt = (next(v0.sc_cycle) if v0.sc_cycle else v0.sc_dir, next(v0.lr_cycle) if v0.lr_cycle else v0.lr_dir)
            ahead = PREPARE_AHEAD and nxt != (v0.sc_dir, v0.lr_dir) and it + 1 < v0.maxit
            norms = np.atleast_1d(dev.cycle(v0.sc_dir, v0.lr_dir, nxt=nxt if ahead else None))
            it += 1
            changed = False
            for b, var in enumerate(vars_):
                if not active[b]:
                    continue
                l2_last[b] = norms[b]
                var.it += 1
                _print_cycle_info(var, l2_last[b], l2_prev[b])
                if _terminate(var, l2_last[b], l2_stag[b][(it - 1) % v0._maxcycle], it):
                    var.l2 = l2_last[b]
                    active[b] = 0
                    changed = True
            # the rotation of the directions depends on the cycle count only: one state for all systems
            v0.sc_dir, v0.lr_dir = nxt
            if changed and active.any():
                dev.set_mask(active)
        efields = [] if download else None
        resp = [] if rec is not None else None
        for b in range(n):
            dev.select(b)
            if rec is not None:
                resp.append(dev.get_receiver_response(rec))
            if download:
                e = fields.Field(grid, dtype=proto.dtype, freq=proto._freq)
                dev.get_efield(np.asarray(e.field))
                efields.append(e)
    finally:
        if handle is None:
            dev.close()
    infos = []
    for var in vars_:
        if var.verb == 1 and var.exit_message != 'CONVERGED':
            var.cprint(f"* WARNING :: {var.exit_message}", 0)
        infos.append({
            'exit': int(var.exit_message != 'CONVERGED'), 'exit_message': var.exit_message, 'abs_error': var.l2,
            'rel_error': var.l2 / var.l2_refe, 'ref_error': var.l2_refe, 'tol': var.tol, 'it_mg': var.it,
            'it_ssl': var._ssl_it, 'time': var.runtime_at_cycle[-1], 'runtime_at_cycle': var.runtime_at_cycle,
            'error_at_cycle': var.error_at_cycle, 'log': var.log_message,
        })
    if rec is not None:
        return efields, infos, np.array(resp)
    return efields, infos


# --------------------------------------------------------------------------
# multigrid / krylov
# --------------------------------------------------------------------------
def multigrid(grid, model, sfield, efield, var, dev=None, **kwargs):
    """Level-0 loop of the multigrid solver (reference emg3d/solver.py:434-607).

    The coarse-level recursion (V/W/F scheduling, restriction, prolongation,
    smoothing) of one cycle runs inside ``emg3d_mg_cycle`` on the device; this
    function owns what the reference does once per cycle on the finest grid:
    initial residual, optional initial smoothing, the end-of-cycle norm,
    sc_dir/lr_dir rotation, logging and the termination tests.  ``efield`` is
    updated in place.
    """
    if kwargs:
        raise TypeError("the coarse-level recursion lives on the device; `level`/`new_cycmax` "
                        "are not accepted here")
    own = dev is None
    if own:
        dev = DeviceMG(grid, model, sfield.dtype)
        dev.set_params(var)
        dev.set_sfield(sfield)
    try:
        if own or not var._dev_efield_current:
            dev.set_efield(efield)
        dev.begin(var.sc_dir)
        it = 0
        l2_last = dev.residual_norm()
        l2_stag = np.ones(var._maxcycle) * l2_last

        # verb = 5 (solver.py:498-515): the norm after every smoothing call of every level -- the device reports them
        tracing = var.verb > 4 and dev.nsys == 1
        cm0 = 1 if var.clevel[var.sc_dir] == 0 else var.cycmax      # level 0's cycmax, fixed on entry (solver.py:480-485)
        var.cprint("     it cycmax               error", 4)
        var.cprint("      level [  dimension  ]            info\n", 4)
        if tracing:
            var.cprint(_print_gs_info(it, 0, cm0, grid.vnC, l2_last) + "initial error", 4)
            dev.set_trace(True)

        if var.nu_init > 0:
            dev.smooth(var.nu_init, var.lr_dir)
            if tracing:
                var.cprint(_print_gs_info(it, 0, cm0, grid.vnC, dev.residual_norm()) + "initial smoothing", 4)

        if var._first_cycle and var.verb > 3:
            var._level_all = _first_cycle_levels(var)       # with the sc_dir of the first cycle

        while True:
            l2_prev = l2_last
            l2_stag[(it - 1) % var._maxcycle] = l2_last

            # the rotation of the directions (solver.py:597-600) is known before the cycle: the device handle prepares
            # the next pair's hierarchy / factorisations / launch graph on the host while the device runs this cycle
            nxt = (next(var.sc_cycle) if var.sc_cycle else var.sc_dir, next(var.lr_cycle) if var.lr_cycle else var.lr_dir)
            ahead = PREPARE_AHEAD and nxt != (var.sc_dir, var.lr_dir) and it + 1 < var.maxit
            l2_last = dev.cycle(var.sc_dir, var.lr_dir, nxt=nxt if ahead else None)
            if tracing:
                for rit, level, cm, kind, vnC, norm in dev.get_trace():
                    var.cprint(_print_gs_info(it if level == 0 else rit, level, cm, vnC, norm) +
                               ("coarsest level", "pre-smoothing", "post-smoothing")[kind], 4)

            it += 1
            var.it += 1
            _print_cycle_info(var, l2_last, l2_prev)

            var.sc_dir, var.lr_dir = nxt

            if _terminate(var, l2_last, l2_stag[(it - 1) % var._maxcycle], it):
                break
        var.l2 = l2_last
        if efield is not None:          # None: the caller picks the field up on the device
            dev.get_efield(np.asarray(efield))
    finally:
        if own:
            dev.close()
        elif var.verb > 4 and dev.nsys == 1:
            dev.set_trace(False)


# BiCGSTAB with every vector resident on the device.  The reference hands this iteration to
# scipy.sparse.linalg.bicgstab (call site emg3d/solver.py:717-719; SciPy is a third-party
# dependency pinned only as scipy>=1.4.0, setup.py:39).  This is a restatement of that
# routine's published algorithm (SciPy 1.12+: scipy/sparse/linalg/_isolve/iterative.py,
# `bicgstab`: same order of operations, same breakdown tests rhotol = omegatol = eps**2, same
# exit codes 0 / maxiter / -10 / -11, atol = max(atol, rtol*||b||)), with numpy operations
# replaced by emg3d_mg_vec_* calls; pinned by the reference's `res>bicresult` and
# `lap>bicresult` goldens and by the host-SciPy path of this module (tests).
DEVICE_KRYLOV = True
# multigrid() / solve_sources(): set the next cycle's (sc_dir, lr_dir) pair up on the host while the device runs the
# current cycle (emg3d_mg_cycle_next); False: set-up on first use, as emg3d_mg_cycle does.  Same results either way.
PREPARE_AHEAD = True


def _bicgstab_device(dev, b, x0, rtol, maxiter, atol, psolve, callback):
    X, R, RT, P, V, S, T, PH, SH, B, TMP, RES = range(12)
    dev.vec_alloc(12)
    dev.vec_set(B, b)
    dev.vec_set(X, x0)
    bnrm2 = dev.vec_norm(B)
    atol = max(float(atol), float(rtol) * float(bnrm2))
    if bnrm2 == 0:
        return np.array(b), 0
    rhotol = np.finfo(dev.dtype.char).eps ** 2
    omegatol = rhotol

    def residual_into(dst):          # dst = b - A x
        dev.vec_amatvec(TMP, X)
        dev.vec_copy(dst, B)
        dev.vec_axpy(dst, -1.0, TMP)

    if np.any(x0):
        residual_into(R)
    else:
        dev.vec_copy(R, B)
    dev.vec_copy(RT, R)
    rho_prev = omega = alpha = None

    def apply_psolve(src, dst):
        if psolve is None:
            dev.vec_copy(dst, src)
        else:
            psolve(src, dst)

    for iteration in range(maxiter):
        if dev.vec_norm(R) < atol:
            return dev.vec_get(X), 0
        rho = dev.vec_dot(RT, R)
        if abs(rho) < rhotol:
            return dev.vec_get(X), -10
        if iteration > 0:
            if abs(omega) < omegatol:
                return dev.vec_get(X), -11
            beta = (rho / rho_prev) * (alpha / omega)
            dev.vec_axpy(P, -omega, V)       # p -= omega*v
            dev.vec_scale(P, beta)           # p *= beta
            dev.vec_axpy(P, 1.0, R)          # p += r
        else:
            dev.vec_copy(P, R)
        apply_psolve(P, PH)
        dev.vec_amatvec(V, PH)
        rv = dev.vec_dot(RT, V)
        if rv == 0:
            return dev.vec_get(X), -11
        alpha = rho / rv
        dev.vec_axpy(R, -alpha, V)           # r -= alpha*v
        dev.vec_copy(S, R)                   # s = r
        if dev.vec_norm(S) < atol:
            dev.vec_axpy(X, alpha, PH)
            return dev.vec_get(X), 0
        apply_psolve(S, SH)
        dev.vec_amatvec(T, SH)
        omega = dev.vec_dot(T, S) / dev.vec_dot(T, T)
        dev.vec_axpy(X, alpha, PH)
        dev.vec_axpy(X, omega, SH)
        dev.vec_axpy(R, -omega, T)
        rho_prev = rho
        if callback:
            residual_into(RES)               # the reference's callback: || sfield - A x ||
            callback(dev.vec_norm(RES))
    return dev.vec_get(X), maxiter


def _cgs_device(dev, b, x0, rtol, maxiter, atol, psolve, callback):
    """SciPy's ``cgs`` (the reference's call site emg3d/solver.py:717-719; SciPy >= 1.12 ``_isolve/iterative.py``)
    restated operation by operation on ``emg3d_mg_vec_*``: every Krylov vector stays in HBM, the host sees two dot
    products and one norm per iteration.  Same breakdown tests and exit codes."""
    X, R, RT, P, U, Q, PH, VH, UQ, UH, B, TMP = range(12)
    dev.vec_alloc(12)
    dev.vec_set(B, b)
    dev.vec_set(X, x0)
    bnrm2 = dev.vec_norm(B)
    atol = max(float(atol), float(rtol) * float(bnrm2))
    if bnrm2 == 0:
        return np.array(b), 0
    rhotol = np.finfo(dev.dtype.char).eps ** 2

    def residual_into(dst):          # dst = b - A x
        dev.vec_amatvec(TMP, X)
        dev.vec_copy(dst, B)
        dev.vec_axpy(dst, -1.0, TMP)

    def apply_psolve(src, dst):
        if psolve is None:
            dev.vec_copy(dst, src)
        else:
            psolve(src, dst)

    if np.any(x0):
        residual_into(R)
    else:
        dev.vec_copy(R, B)
    dev.vec_copy(RT, R)
    rho_prev = None
    for iteration in range(maxiter):
        if dev.vec_norm(R) < atol:
            return dev.vec_get(X), 0
        rho = dev.vec_dot(RT, R)
        if abs(rho) < rhotol:
            return dev.vec_get(X), -10
        if iteration > 0:
            beta = rho / rho_prev
            dev.vec_copy(U, R)               # u = r + beta q
            dev.vec_axpy(U, beta, Q)
            dev.vec_scale(P, beta)           # p = u + beta (q + beta p)
            dev.vec_axpy(P, 1.0, Q)
            dev.vec_scale(P, beta)
            dev.vec_axpy(P, 1.0, U)
        else:
            dev.vec_copy(P, R)
            dev.vec_copy(U, R)
        apply_psolve(P, PH)
        dev.vec_amatvec(VH, PH)
        rv = dev.vec_dot(RT, VH)
        if rv == 0:
            return dev.vec_get(X), -11
        alpha = rho / rv
        dev.vec_copy(Q, U)                   # q = u - alpha vhat
        dev.vec_axpy(Q, -alpha, VH)
        dev.vec_copy(UQ, U)                  # uhat = M (u + q)
        dev.vec_axpy(UQ, 1.0, Q)
        apply_psolve(UQ, UH)
        dev.vec_axpy(X, alpha, UH)
        residual_into(R)                     # the true residual, as SciPy computes it
        rho_prev = rho
        if callback:
            callback(dev.vec_norm(R))        # the reference's callback: || sfield - A x ||
    return dev.vec_get(X), maxiter


def _krylov_fits_device(dev, name, m=20):
    """Do the Krylov vectors of the device-resident iteration fit next to the handle?  GCROT(m, k = m) keeps up to about
    5 + 2 (m + 1) + 2 m + 2 nE-sized vectors in HBM (~1.5 GB each at 256^3: ~130 GB), bicgstab / cgs 9 / 10.  When
    they do not fit, the caller runs SciPy's host iteration around the device preconditioner instead of failing in
    ``emg3d_mg_vec_alloc``."""
    nvec = {'bicgstab': 9, 'cgs': 10}.get(name, 5 + 2 * (m + 1) + 2 * m + 2)
    need = nvec * dev.nE * dev.dtype.itemsize
    try:
        # what is FREE now (other handles, torch allocations and other processes share the GPU), plus the DEVICE blocks this
        # process's own pool has parked for this device, which the vectors may take (not the blocks of other devices, not the
        # pinned host staging buffers)
        mi = _lib.mem_info(dev.device)
    except Exception:
        return True
    return need < 0.92 * (mi["free"] + mi["pooled_on_device"])


def _gcrotmk_device(dev, b, x0, rtol, maxiter, atol, psolve, callback, m=20, k=None):
    """SciPy's ``gcrotmk`` (GCROT(m,k) with its flexible inner GMRES ``_fgmres``; the reference's call site
    emg3d/solver.py:717-719 with SciPy's defaults m = 20, k = m, truncate = 'oldest', no recycled vectors;
    ``_isolve/_gcrotmk.py``) restated operation by operation on ``emg3d_mg_vec_*``: the Krylov, preconditioned and
    outer (C, U) vectors stay in HBM -- allocated as they are needed, recycled between outer iterations --, the host sees
    the dot products and keeps the small matrices (Hessenberg QR by ``scipy.linalg.qr_insert``, ``lstsq``) as SciPy does.
    Same breakdown tests and exit codes."""
    from scipy.linalg import qr_insert, lstsq
    if k is None:
        k = m
    dtype = dev.dtype
    free, top = [], [0]

    def new_vec():
        if free:
            return free.pop()
        top[0] += 1
        dev.vec_alloc(top[0])
        return top[0] - 1

    X, R, B, TMP, RT = new_vec(), new_vec(), new_vec(), new_vec(), new_vec()
    dev.vec_set(B, b)
    dev.vec_set(X, x0)

    def residual_into(dst):          # dst = b - A x
        dev.vec_amatvec(TMP, X)
        dev.vec_copy(dst, B)
        dev.vec_axpy(dst, -1.0, TMP)

    if np.any(x0):
        residual_into(R)
    else:
        dev.vec_copy(R, B)
    b_norm = dev.vec_norm(B)
    atol = max(float(atol), float(rtol) * float(b_norm))
    if b_norm == 0:
        return np.array(b), 0
    eps = np.finfo(dtype).eps
    CU = []                          # [(c, u)] vector ids, oldest first

    def fgmres(V0, ml, atol_in, cs):
        """_fgmres with right preconditioning; V0 is normalised.  Returns Q, R, B, vs, zs, y (ids in vs / zs)."""
        vs, zs = [V0], []
        Bm = np.zeros((len(cs), ml), dtype=dtype)
        Q = np.ones((1, 1), dtype=dtype)
        Rm = np.zeros((1, 0), dtype=dtype)
        breakdown = False
        j = 0
        for j in range(ml):
            z = new_vec()
            if psolve is None:
                dev.vec_copy(z, vs[-1])
            else:
                psolve(vs[-1], z)
            w = new_vec()
            dev.vec_amatvec(w, z)
            w_norm = dev.vec_norm(w)
            for i, c in enumerate(cs):              # GCROT projection: orthogonalise against C
                alpha = dev.vec_dot(c, w)
                Bm[i, j] = alpha
                dev.vec_axpy(w, -alpha, c)
            hcur = np.zeros(j + 2, dtype=Q.dtype)
            for i, v in enumerate(vs):              # ... against V
                alpha = dev.vec_dot(v, w)
                hcur[i] = alpha
                dev.vec_axpy(w, -alpha, v)
            hcur[len(vs)] = dev.vec_norm(w)
            with np.errstate(over='ignore', divide='ignore'):
                alpha = 1 / hcur[-1]
            if np.isfinite(alpha):
                dev.vec_scale(w, alpha)
            if not (hcur[-1] > eps * w_norm):
                breakdown = True
            vs.append(w)
            zs.append(z)
            Q2 = np.zeros((j + 2, j + 2), dtype=Q.dtype, order='F')
            Q2[:j + 1, :j + 1] = Q
            Q2[j + 1, j + 1] = 1
            R2 = np.zeros((j + 2, j), dtype=Rm.dtype, order='F')
            R2[:j + 1, :] = Rm
            Q, Rm = qr_insert(Q2, R2, hcur, j, which='col', overwrite_qru=True, check_finite=False)
            res = abs(Q[0, -1])
            if res < atol_in or breakdown:
                break
        if not np.isfinite(Rm[j, j]):
            raise np.linalg.LinAlgError()
        y, _, _, _ = lstsq(Rm[:j + 1, :j + 1], Q[0, :j + 1].conj())
        return Q, Rm, Bm[:, :j + 1], vs, zs, y

    j_outer = -1
    for j_outer in range(maxiter):
        if callback is not None:
            residual_into(RT)
            callback(dev.vec_norm(RT))              # the reference's callback: || sfield - A x ||
        beta = dev.vec_norm(R)
        beta_tol = max(atol, rtol * b_norm)
        if beta <= beta_tol and (j_outer > 0 or CU):
            residual_into(R)                        # recompute the residual to avoid rounding error
            beta = dev.vec_norm(R)
        if beta <= beta_tol:
            j_outer = -1
            break
        ml = m + max(k - len(CU), 0)
        cs = [c for c, u in CU]
        V0 = new_vec()
        dev.vec_copy(V0, R)
        dev.vec_scale(V0, 1 / beta)
        try:
            Q, Rm, Bm, vs, zs, y = fgmres(V0, ml, max(atol, rtol * b_norm) / beta, cs)
            y = y * beta
        except np.linalg.LinAlgError:
            break
        # ux := (Z - U B) y
        ux = new_vec()
        dev.vec_copy(ux, zs[0])
        dev.vec_scale(ux, y[0])
        for z, yc in zip(zs[1:], y[1:]):
            dev.vec_axpy(ux, yc, z)
        by = Bm.dot(y)
        for (c, u), byc in zip(CU, by):
            dev.vec_axpy(ux, -byc, u)
        # cx := V H y
        with np.errstate(invalid="ignore"):
            hy = Q.dot(Rm.dot(y))
        cx = new_vec()
        dev.vec_copy(cx, vs[0])
        dev.vec_scale(cx, hy[0])
        for v, hyc in zip(vs[1:], hy[1:]):
            dev.vec_axpy(cx, hyc, v)
        free.extend(vs)                             # the inner vectors are done with
        free.extend(zs)
        try:
            with np.errstate(divide='raise', invalid='raise'):
                alpha = 1 / dev.vec_norm(cx)
            if not np.isfinite(alpha):
                raise FloatingPointError()
        except (FloatingPointError, ZeroDivisionError):
            free.extend([cx, ux])
            continue
        dev.vec_scale(cx, alpha)
        dev.vec_scale(ux, alpha)
        gamma = dev.vec_dot(cx, R)
        dev.vec_axpy(R, -gamma, cx)
        dev.vec_axpy(X, gamma, ux)
        while len(CU) >= k and CU:                  # truncate = 'oldest'
            c, u = CU.pop(0)
            free.extend([c, u])
        CU.append((cx, ux))
    else:
        return dev.vec_get(X), maxiter
    return dev.vec_get(X), j_outer + 1


def krylov(grid, model, sfield, efield, var, dev=None):
    """Krylov solver preconditioned by multigrid (reference solver.py:610-734).

    ``bicgstab``, ``cgs`` and ``gcrotmk`` run device resident (``_bicgstab_device``, ``_cgs_device``, ``_gcrotmk_device``:
    SciPy's iterations restated on vectors in HBM; gcrotmk's Hessenberg QR / least-squares bookkeeping on small matrices
    stays host work, as in SciPy).  ``solver.DEVICE_KRYLOV = False`` selects SciPy's own iterations on host vectors with
    the operator A x (``core.amat_x``) and the preconditioner (multigrid cycles on a zero field) on the device: 2 x nE x
    16 B cross PCIe per operator or preconditioner application.
    """
    own = dev is None
    if own:
        dev = DeviceMG(grid, model, sfield.dtype)
        dev.set_params(var)
    freq = sfield._freq

    def amatvec(x):
        return dev.amatvec(np.asarray(x))

    A = ssl.LinearOperator(shape=(grid.nE, grid.nE), dtype=sfield.dtype, matvec=amatvec)

    def mg_matvec(b):
        bs = fields.Field(grid, np.ascontiguousarray(b, dtype=sfield.dtype), freq=freq)
        x = fields.Field(grid, dtype=sfield.dtype, freq=freq)
        dev.set_sfield(bs)
        dev.set_efield(None)
        var._dev_efield_current = True
        try:
            multigrid(grid, model, bs, x, var, dev=dev)
        finally:
            var._dev_efield_current = False
        return x

    M = None
    if var.cycle:
        M = ssl.LinearOperator(shape=(grid.nE, grid.nE), dtype=sfield.dtype, matvec=mg_matvec)

    def callback(x):
        var._ssl_it += 1
        var.runtime_at_cycle = np.r_[var.runtime_at_cycle, var.time.elapsed]
        if isinstance(x, float):        # device path: the norm of s - A x, computed on the device
            var.l2 = x
        else:
            r = np.asarray(sfield) - dev.amatvec(np.asarray(x))
            var.l2 = float(np.linalg.norm(r))
        var.error_at_cycle = np.r_[var.error_at_cycle, var.l2]
        if var.verb > 3:
            log = f"   [{var.time.now}]   {var.l2/var.l2_refe:.3e} "
            log += f" after {var._ssl_it:3} {var.sslsolver}-cycles"
            if var._ssl_it == 1 and var.it == 0 and var.cycle is not None:
                log += "\n"
            var.cprint(log, 3)
        elif var.verb < 0:
            var.one_liner(var.l2)

    def mg_on_device(src, dst):
        """dst = M src with both vectors on the device (same cycles as mg_matvec)."""
        dev.vec_copy(dev.SFIELD, src)
        dev.set_efield(None)
        var._dev_efield_current = True
        try:
            multigrid(grid, model, None, None, var, dev=dev)
        finally:
            var._dev_efield_current = False
        dev.vec_copy(dst, dev.EFIELD)

    try:
        if var.sslsolver in ('bicgstab', 'cgs', 'gcrotmk') and DEVICE_KRYLOV and _krylov_fits_device(dev, var.sslsolver):
            drive = {'bicgstab': _bicgstab_device, 'cgs': _cgs_device, 'gcrotmk': _gcrotmk_device}[var.sslsolver]
            x, i = drive(dev, np.asarray(sfield), np.asarray(efield), rtol=var.tol, maxiter=var.ssl_maxit, atol=1e-30,
                         psolve=mg_on_device if var.cycle else None, callback=callback)
        else:
            x, i = getattr(ssl, var.sslsolver)(A, np.asarray(sfield), x0=np.array(efield), rtol=var.tol,
                                              maxiter=var.ssl_maxit, atol=1e-30, M=M, callback=callback)
        efield.field = x
    except _ConvergenceError:
        i = -1
        var.exit_message += " (returned field is zero)"
    finally:
        if own:
            dev.close()

    pre = "\n   > "
    if i < 0:
        if var.exit_message == '':
            var.exit_message = f"Error in {var.sslsolver} ({i})"
        pre = "\n* ERROR   :: "
    elif i > 0:
        var.exit_message = "MAX. ITERATION REACHED, NOT CONVERGED"
    else:
        var.exit_message = "CONVERGED"
    var.cprint(pre + var.exit_message, 2)


# --------------------------------------------------------------------------
# Stand-alone sub-routines on host arrays (tier 1)
# --------------------------------------------------------------------------
def smoothing(grid, model, sfield, efield, nu, lr_dir, ordering='lex'):
    """Gauss-Seidel smoothing, in place (reference emg3d/solver.py:738-799)."""
    inp = (sfield.fx, sfield.fy, sfield.fz, model.eta_x, model.eta_y, model.eta_z, model.zeta,
           grid.h[0], grid.h[1], grid.h[2], nu)
    lr_dir = _current_lr_dir(lr_dir, grid)
    order = ORDERINGS[ordering]
    if lr_dir == 0:
        core._gs(0, efield.fx, efield.fy, efield.fz, *inp, order=order)
    if lr_dir in [1, 5, 6, 7]:
        core._gs(1, efield.fx, efield.fy, efield.fz, *inp, order=order)
    if lr_dir in [2, 4, 6, 7]:
        core._gs(2, efield.fx, efield.fy, efield.fz, *inp, order=order)
    if lr_dir in [3, 4, 5, 7]:
        core._gs(3, efield.fx, efield.fy, efield.fz, *inp, order=order)


class _CoarseModel:
    """Coarse-grid VolumeModel stand-in (eta_x/y/z, zeta, case)."""

    def __init__(self, case):
        self.case = case


def _restrict_model_parameters(param, sc_dir):
    """Coarse model parameter = sum of the merged fine cells (solver.py:1747-1784)."""
    lib = _lib.load()
    param = np.asarray(param)
    is_c = int(param.dtype == np.complex128)
    dt = np.complex128 if is_c else np.float64
    p = np.ascontiguousarray(np.asarray(param, dtype=dt).ravel(order='F'))
    nx, ny, nz = param.shape
    cshape = (nx if sc_dir in [1, 5, 6] else nx // 2, ny if sc_dir in [2, 4, 6] else ny // 2,
              nz if sc_dir in [3, 4, 5] else nz // 2)
    out = np.empty(int(np.prod(cshape)), dtype=dt)
    _lib.check(lib.emg3d_restrict_model(is_c, nx, ny, nz, _lib.ptr(out), _lib.ptr(p), int(sc_dir)),
               "emg3d_restrict_model")
    return out.reshape(cshape, order='F')


def restriction(grid, model, sfield, residual, sc_dir):
    """Coarse grid, model and source from the fine ones (solver.py:802-901)."""
    rx = 1 if sc_dir in [1, 5, 6] else 2
    ry = 1 if sc_dir in [2, 4, 6] else 2
    rz = 1 if sc_dir in [3, 4, 5] else 2
    ch = [np.diff(grid.nodes_x[::rx]), np.diff(grid.nodes_y[::ry]), np.diff(grid.nodes_z[::rz])]
    cgrid = meshes.TensorMesh(ch, grid.origin)

    cmodel = _CoarseModel(model.case)
    cmodel.eta_x = _restrict_model_parameters(model.eta_x, sc_dir)
    cmodel.eta_y = _restrict_model_parameters(model.eta_y, sc_dir) if model.case in [1, 3] else cmodel.eta_x
    cmodel.eta_z = _restrict_model_parameters(model.eta_z, sc_dir) if model.case in [2, 3] else cmodel.eta_x
    cmodel.zeta = _restrict_model_parameters(model.zeta, sc_dir)

    wx, wy, wz = _get_restriction_weights(grid, cgrid, sc_dir)
    csfield = fields.Field(cgrid, dtype=sfield.dtype, freq=sfield._freq)
    core.restrict(csfield.fx, csfield.fy, csfield.fz, residual.fx, residual.fy, residual.fz,
                  wx, wy, wz, sc_dir)
    csfield.ensure_pec
    cefield = fields.Field(cgrid, dtype=sfield.dtype, freq=sfield._freq)
    return cgrid, cmodel, csfield, cefield


def prolongation(grid, efield, cgrid, cefield, sc_dir):
    """efield += P cefield, then PEC (reference emg3d/solver.py:904-977)."""
    lib = _lib.load()
    dt = _lib.dtype_code(efield.dtype)
    hx, hy, hz = (np.ascontiguousarray(h, dtype=np.float64) for h in grid.h)
    origin = np.ascontiguousarray(grid.origin, dtype=np.float64)
    e = np.ascontiguousarray(np.asarray(efield))
    ce = np.ascontiguousarray(np.asarray(cefield), dtype=e.dtype)
    _lib.check(lib.emg3d_prolongation(dt, *(int(n) for n in grid.vnC), _lib.ptr(hx), _lib.ptr(hy),
                                      _lib.ptr(hz), _lib.ptr(origin), _lib.ptr(e), _lib.ptr(ce),
                                      int(sc_dir)), "emg3d_prolongation")
    if e is not np.asarray(efield):
        efield.field = e
    else:
        np.asarray(efield)[...] = e


def residual(grid, model, sfield, efield, norm=False):
    """Residual field s - A e, or its l2-norm (reference solver.py:980-1039)."""
    rfield = sfield.copy()
    core.amat_x(rfield.fx, rfield.fy, rfield.fz, efield.fx, efield.fy, efield.fz, model.eta_x,
                model.eta_y, model.eta_z, model.zeta, grid.h[0], grid.h[1], grid.h[2])
    if norm:
        return float(np.linalg.norm(rfield))
    return rfield


# --------------------------------------------------------------------------
# Parameters
# --------------------------------------------------------------------------
class _Time:
    """Wall-clock helper with the reference's formatting (utils.py:604-634)."""

    def __init__(self):
        self._t0 = time.perf_counter()

    @property
    def now(self):
        return datetime.now().strftime("%H:%M:%S")

    @property
    def elapsed(self):
        return time.perf_counter() - self._t0

    @property
    def runtime(self):
        return str(timedelta(seconds=np.round(self.elapsed)))


@dataclass
class MGParameters:
    """Multigrid settings; mirrors emg3d/solver.py:1043-1364."""

    verb: int
    cycle: str
    sslsolver: str
    linerelaxation: int
    semicoarsening: int
    vnC: tuple

    tol: float = 1e-6
    maxit: int = 50
    nu_init: int = 0
    nu_pre: int = 2
    nu_coarse: int = 1
    nu_post: int = 2
    clevel: int = -1
    return_info: bool = False
    log: int = 1
    log_message: str = ''
    ordering: str = 'colour'

    def __post_init__(self):
        if self.ordering not in ORDERINGS:
            raise ValueError(f"`ordering` must be one of {list(ORDERINGS)}; provided: {self.ordering!r}.")
        self._level_all = list()
        self._first_cycle = True
        self._dev_efield_current = False
        self.it = 0
        self._ssl_it = 0
        self.l2 = 1.0
        self.l2_refe = 1.0
        self.exit_message = ''
        self.time = _Time()
        self.runtime_at_cycle = np.array([0.])
        self.error_at_cycle = np.array([0.])
        self.do_return = True
        self._semicoarsening()
        self._linerelaxation()
        self._solver_and_cycle()
        self.max_level

    def __repr__(self):
        n = self.vnC
        p = self.pclevel
        return (f"   MG-cycle       : {self.cycle!r:17}   sslsolver : {self.sslsolver!r}\n"
                f"   semicoarsening : {self._p_sc_dir:17}   tol       : {self.tol}\n"
                f"   linerelaxation : {self._p_lr_dir:17}   maxit     : {self._maxit}\n"
                f"   nu_{{i,1,c,2}}   : {self.nu_init}, {self.nu_pre}, {self.nu_coarse}, "
                f"{self.nu_post}          verb      : {self.verb}\n"
                f"   Original grid  : {n[0]:3} x {n[1]:3} x {n[2]:3}     => {n[0]*n[1]*n[2]:,} cells\n"
                f"   Coarsest grid  : {p['vnC'][0]:3} x {p['vnC'][1]:3} x {p['vnC'][2]:3}     "
                f"=> {p['nC']:,} cells\n"
                f"   Coarsest level : {p['clevel'][0]:3} ; {p['clevel'][1]:3} ;{p['clevel'][2]:4}   "
                f"{p['message']}\n"
                f"   ordering       : {self.ordering}\n")

    @staticmethod
    def _halvings(n):
        """How often ``n`` cells can be halved with at least two cells left: the trailing zero bits of ``n``, one fewer for a
        pure power of two (2^k stops at two cells)."""
        n = int(n)
        if n < 2:
            return 0
        tz = (n & -n).bit_length() - 1
        return tz - 1 if n >> tz == 1 else tz

    @property
    def max_level(self):
        """Coarsening depth per dimension, limited by the user's ``clevel``; sets ``clevel`` (the depth for sc_dir 0..3),
        ``pclevel`` (coarsest grid, for the log) and the "not optimal" note (behaviour of solver.py:1142-1206)."""
        asked = self.clevel
        cap = np.inf if asked < 0 else asked
        depth = np.array([self._halvings(n) for n in self.vnC], dtype=np.int_)
        if asked >= 0:
            depth = np.minimum(depth, asked).astype(np.int_)
        coarse = tuple(int(n) >> int(d) for n, d in zip(self.vnC, depth))
        # sc_dir d leaves dimension d alone: its depth is the deepest of the other two (sc_dir 0: of all three)
        self.clevel = np.array([depth.max()] + [max(depth[k] for k in range(3) if k != d) for d in range(3)])
        stuck_large = any(d < cap and c > 7 for d, c in zip(depth, coarse))      # stopped by an odd factor with > 7 cells left
        shallow = bool(np.any(depth < min(cap, 3)))                               # fewer than three halvings somewhere
        self.pclevel = {'nC': int(np.prod(coarse)), 'vnC': coarse, 'clevel': depth,
                        'message': "  :: Grid not optimal for MG solver ::" if (stuck_large or shallow) else ""}
        if min(self.vnC) < 2:
            raise ValueError("Nr. of cells must be at least two in each direction\n"
                             f"Provided shape: ({self.vnC[0]}, {self.vnC[1]}, {self.vnC[2]}).")

    def cprint(self, info, verbosity, **kwargs):
        """Print/log ``info`` if ``verb`` > ``verbosity`` (solver.py:1208-1229)."""
        if self.verb > verbosity:
            if self.log != 0:
                self.log_message += str(info) + '\n'
            if self.log >= 0:
                print(info, **kwargs)

    def one_liner(self, l2_last, last=False):
        info = f":: emg3d :: {l2_last/self.l2_refe:.1e}; "
        info += f"{self._ssl_it}({self.it}); " if self.sslsolver else f"{self.it}; "
        info += f"{self.time.runtime}"
        if last:
            self.cprint(info + f"; {self.exit_message}", -100)
        else:
            self.cprint(info, -100, end='\r')

    def _digits(self, value, true_cycle, nmax, name, allowed, combo):
        if value is True:
            raw = np.array(true_cycle)
            return itertools.cycle(raw), raw
        if value in np.arange(nmax):
            return False, np.array([int(value)])
        raw = np.array([int(x) for x in str(abs(value))])
        if np.any(raw < 0) or np.any(raw >= nmax):
            raise ValueError(f"`{name}` must be one of {allowed}.\n"
                             f"{' ':>13} Or a combination of {combo} to cycle, e.g. 1213.\n"
                             f"{'Provided:':>23} {name}={value}.")
        return itertools.cycle(raw), raw

    def _semicoarsening(self):
        self.sc_cycle, raw = self._digits(self.semicoarsening, [1, 2, 3], 4, 'semicoarsening',
                                          "(False, True, 0, 1, 2, 3)", "(0, 1, 2, 3)")
        self.sc_dir = next(self.sc_cycle) if self.sc_cycle else raw[0]
        self.semicoarsening = self.sc_dir != 0
        self._p_sc_dir = f"{self.semicoarsening} {raw}"
        self._raw_sc_cycle = raw

    def _linerelaxation(self):
        self.lr_cycle, raw = self._digits(self.linerelaxation, [4, 5, 6], 8, 'linerelaxation',
                                          "(False, True, 0, 1, 2, 3, 4, 5, 6, 7)", "(1, 2, 3, 4, 5, 6, 7)")
        self.lr_dir = next(self.lr_cycle) if self.lr_cycle else raw[0]
        self.linerelaxation = self.lr_dir != 0
        self._p_lr_dir = f"{self.linerelaxation} {raw}"
        self._raw_lr_cycle = raw

    def _solver_and_cycle(self):
        """Validate the (sslsolver, cycle) pair and derive cycmax / the iteration limits (behaviour of solver.py:1327-1364)."""
        known = ['bicgstab', 'cgs', 'gcrotmk']
        given = self.sslsolver
        if given is not True and given is not False and given not in known:
            raise ValueError(f"`sslsolver` must be True, False, or one of {known}.\n"
                             f"Provided: sslsolver={given!r}.")
        self.sslsolver = known[0] if given is True else given
        if self.cycle not in ('F', 'V', 'W', None):
            raise ValueError("`cycle` must be one of {'F', 'V', 'W', None}.\n"
                             f"Provided: cycle={self.cycle}.")
        if self.cycle is None and not self.sslsolver:
            raise ValueError("At least `cycle` or `sslsolver` is required.\nProvided"
                             f"input: cycle={self.cycle}; sslsolver={self.sslsolver}.")
        self.cycmax = {'F': 2, 'W': 2}.get(self.cycle, 1)
        # one multigrid iteration per entry of the longer of the two rotation lists when it preconditions a Krylov solver
        self._maxcycle = max(len(self._raw_sc_cycle), len(self._raw_lr_cycle))
        self._maxit = f"{self.maxit}"
        self.ssl_maxit = self.maxit if self.sslsolver else 0
        if self.sslsolver and self.cycle is not None:
            self.maxit = self._maxcycle
            self._maxit += f" ({self.maxit})"


# --------------------------------------------------------------------------
# Helpers
# --------------------------------------------------------------------------
class RegularGridProlongator:
    """Bilinear interpolation from a coarse tensor grid ``(x, y)`` to fixed points ``cxy``
    (interface and semantics of the reference's class, emg3d/solver.py:1368-1463: linear, no bounds
    error, linear EXTRApolation outside, cell index ``searchsorted - 1`` clipped to the last interval).

    The interval index and the four corner weights of every point are computed once; a call gathers the
    four corners of a 2-D value array and adds them in the order (0,0), (0,1), (1,0), (1,1) -- the
    arithmetic that the device kernel ``k_prolong`` reproduces (its weights come from the same rule,
    ``prolong_weights_host`` in csrc/mg.hpp).  Host-side: used by tests and by code written against
    the reference; the product path prolongates on the device.
    """

    def __init__(self, x, y, cxy):
        cxy = np.asarray(cxy, dtype=np.float64)
        self.size = cxy.shape[0]
        idx, frac = [], []
        for pts, nodes in zip(cxy.T, (np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))):
            i = np.clip(np.searchsorted(nodes, pts) - 1, 0, nodes.size - 2)
            idx.append(i)
            frac.append((pts - nodes[i]) / (nodes[i + 1] - nodes[i]))
        (ix, iy), (tx, ty) = idx, frac
        self._corners = ((ix, iy), (ix, iy + 1), (ix + 1, iy), (ix + 1, iy + 1))
        self.weight = np.array([(1 - tx) * (1 - ty), (1 - tx) * ty, tx * (1 - ty), tx * ty])

    def __call__(self, values):
        values = np.asarray(values)
        result = 0.
        for corner, w in zip(self._corners, self.weight):
            result = result + values[corner] * w
        return result


def _get_prolongation_coordinates(grid, d1, d2):
    """All (d1, d2) node coordinate pairs of ``grid``, d1 running fastest (reference
    emg3d/solver.py:1841-1845): the points at which a coarse field is evaluated for prolongation."""
    n1 = np.asarray(getattr(grid, 'nodes_' + d1), dtype=np.float64)
    n2 = np.asarray(getattr(grid, 'nodes_' + d2), dtype=np.float64)
    out = np.empty((n1.size * n2.size, 2))
    out[:, 0] = np.tile(n1, n2.size)
    out[:, 1] = np.repeat(n2, n1.size)
    return out


def _current_sc_dir(sc_dir, grid):
    """Actual coarsening code 0..6 for this grid (solver.py:1467-1514)."""
    n = grid.vnC
    xs = n[0] % 2 != 0 or n[0] < 3 or sc_dir == 1
    ys = n[1] % 2 != 0 or n[1] < 3 or sc_dir == 2
    zs = n[2] % 2 != 0 or n[2] < 3 or sc_dir == 3
    if xs:
        if ys:
            return 6
        return 5 if zs else 1
    if ys:
        return 4 if zs else 2
    return 3 if zs else 0


def _current_lr_dir(lr_dir, grid):
    """Drop line relaxation along 2-cell dimensions (solver.py:1517-1572)."""
    lr_dir = int(lr_dir)
    n = grid.vnC
    if n[0] == 2:
        lr_dir = {1: 0, 5: 3, 6: 2, 7: 4}.get(lr_dir, lr_dir)
    if n[1] == 2:
        lr_dir = {2: 0, 4: 3, 6: 1, 7: 5}.get(lr_dir, lr_dir)
    if n[2] == 2:
        lr_dir = {3: 0, 4: 2, 5: 1, 7: 6}.get(lr_dir, lr_dir)
    return lr_dir


def _first_cycle_levels(var):
    """The levels in the order ``solver.multigrid`` enters and re-enters them during the FIRST cycle (the reference
    records them for its cycle-QC figure, solver.py:494-496 and 565-567).  The recursion itself runs on the device
    (``MG<T>::mg_level``); this is the same V/W/F rule (solver.py:478-485, 519, 585-586) on level numbers only."""
    coarsest = int(var.clevel[var.sc_dir])
    seq = []

    def visit(level, new_cycmax):
        if level == coarsest:
            cycmax = 1
        elif new_cycmax == 0 or var.cycle != 'F':
            cycmax = var.cycmax
        else:
            cycmax = new_cycmax
        seq.append(level)
        it = cyc = 0
        while level == 0 or it < cycmax:
            if level != coarsest:
                visit(level + 1, cycmax - cyc)
                seq.append(level)
            it += 1
            if level == 0:
                break           # the figure is drawn at the end of the first level-0 iteration
            cyc += 1

    visit(0, 0)
    return seq


def _cycle_qc_figure(levels):
    """ASCII picture of one multigrid cycle, the format of the reference's log at verb > 3 (solver.py:1603-1632): one row
    per coarse level, a backslash where the cycle steps down into that level, a slash where it comes back up; at most
    70 steps."""
    lv = np.asarray(levels, dtype=np.int_)
    top = int(lv.max()) if lv.size else 0
    step = ((lv[1:] + lv[:-1]) // 2 + 1) * (lv[1:] - lv[:-1])      # +k: down into level k, -k: up out of it
    shown = step[:70]
    rows = ["       h_"]
    for k in range(1, top + 1):
        rows.append(f"   {2**k:4}h_ " + "".join("\\" if v == k else "/" if v == -k else " " for v in shown))
    out = "\n".join(rows) + ("\n\n" if top > 0 else "\n\n\n")
    if step.size > 70:
        out += f"  (Cycle-QC restricted to first 70 steps of {step.size} steps.)\n"
    return out


def _print_gs_info(it, level, cycmax, vnC, norm):
    """Info string logged after a smoothing call at verb = 5 (solver.py:1651-1680)."""
    return f"     {it:2} {level} {cycmax} [{vnC[0]:3}, {vnC[1]:3}, {vnC[2]:3}]: {norm:.3e} "


def _print_cycle_info(var, l2_last, l2_prev):
    """Bookkeeping + log line at the end of a cycle (solver.py:1575-1648), with the cycle-QC figure in front of the
    first cycle's line at verb > 3 and a blank line before and after the line at verb > 4."""
    var.runtime_at_cycle = np.r_[var.runtime_at_cycle, var.time.elapsed]
    var.error_at_cycle = np.r_[var.error_at_cycle, l2_last]
    if var.verb < 0:
        var.one_liner(l2_last)
        return
    elif var.verb < 4:
        return
    info = "\n" if var.verb > 4 else ""
    if var._first_cycle:
        info += _cycle_qc_figure(var._level_all or _first_cycle_levels(var))
        var._first_cycle = False
    info += f"   [{var.time.now}]   {l2_last/var.l2_refe:.3e}  "
    if var.sslsolver:
        info += f"after {19*' '} {var.it:3} {var.cycle}-cycles "
    else:
        info += f"after {var.it:3} {var.cycle}-cycles   "
        info += f"[{l2_last:.3e}, {l2_last/l2_prev:.3f}]"
    info += f"   {var.lr_dir} {var.sc_dir}"
    if var.verb > 4:
        info += "\n"
    var.cprint(info, 3)


def _terminate(var, l2_last, l2_stag, it):
    """End of an iteration: converged / diverged / stagnated / out of iterations, tested in that order (behaviour of
    solver.py:1682-1744).  A failing preconditioner aborts the Krylov solver through _ConvergenceError."""
    ref = var.l2_refe
    if l2_last < var.tol * ref:
        verdict, failed = "CONVERGED", False
    elif not np.isfinite(l2_last) or l2_last > 10 * ref:
        verdict, failed = "DIVERGED", True
    elif it > 2 and l2_last >= l2_stag:
        verdict, failed = "STAGNATED", True
    elif it == var.maxit:
        verdict, failed = (None if var.sslsolver else "MAX. ITERATION REACHED, NOT CONVERGED"), False
    else:
        return False
    if verdict is not None:
        var.exit_message = verdict
    if var.sslsolver:
        if failed:
            raise _ConvergenceError
    else:
        var.cprint(("\n" if var.verb < 5 else "") + "   > " + var.exit_message, 2)
    return True


def _get_restriction_weights(grid, cgrid, sc_dir):
    """Restriction weights per axis; dummies for non-coarsened axes
    (solver.py:1787-1838)."""
    out = []
    axes = (('x', [1, 5, 6]), ('y', [2, 4, 6]), ('z', [3, 4, 5]))
    for a, (c, skip) in enumerate(axes):
        if sc_dir not in skip:
            out.append(core.restrict_weights(
                getattr(grid, 'nodes_' + c), getattr(grid, 'cell_centers_' + c), grid.h[a],
                getattr(cgrid, 'nodes_' + c), getattr(cgrid, 'cell_centers_' + c), cgrid.h[a]))
        else:
            wlr = np.zeros(grid.vnN[a], dtype=np.float64)
            w0 = np.ones(grid.vnN[a], dtype=np.float64)
            out.append((wlr, w0, wlr))
    return out


class _ConvergenceError(Exception):
    """Raised inside the preconditioner to abort the SciPy solver."""
    pass
