"""
Generate the golden fixtures under tests/golden/ by IMPORTING the reference
(emg3d v0.17.0 mounted read-only at /root/reference) in the build container.

The reference needs numba, which is not installed; following SURVEY.md App. D
it is imported with a no-op ``numba.njit`` stub (pure-Python loops), three
NumPy-2 aliases and a SciPy ``tol -> rtol`` shim, all created in a temporary
directory (nothing of the reference is written into this repository; the
fixtures hold inputs and expected outputs only).

Run:  python tests/golden/make_golden.py [--big]

Outputs (np.savez_compressed):
  kernels_c128.npz / kernels_f64.npz   per-kernel in/out pairs (core.py + the
                                        solver.py transfer operators)
  regression.npz                        inputs + golden fields of the
                                        reference's tests/data/regression.npz
                                        (res, reg_2, lap) re-exported with
                                        plain keys, plus per-cycle error traces
                                        captured by running the reference here
  solves_16.npz                         16^3 stretched random tri-axial solves
  solves_eps.npz                        the same grid with epsilon_r and mu_r, frequency and Laplace domain (eta arrays + F-cycle solves)
                                        (V/F/W, sc+lr, BiCGSTAB) : traces+fields
  kernels_colour.npz                    the device's 4-/8-colour smoother schedule replayed with the reference's own
                                        core.gauss_seidel* on 2 x 2 (x 2)-cell sub-grids: nu = 1, 2, 3 on
                                        the kernel fixtures' inputs (c128, f64), on a ragged odd grid and on a 70 x 6 x 5 grid (long lines)
  solves_16_colour.npz                  the solves_16 problem solved by the reference's own solver.solve with its smoothing calls
                                        replaced by the colour-schedule replay (reference kernels, the device's order): F / V sc+lr, F plain, BiCGSTAB + F sc+lr
  source_fields.npz                     get_source_field in/out pairs
  gradient.npz                          adjoint-state gradient of one (source, frequency) pair on its computational
                                        grid: the reference's get_source_field / solve / get_receiver_response /
                                        edges2cellaverages composed as Simulation._get_rfield / optimize.gradient do
  receivers.npz                         get_receiver_response (electric + magnetic fields; inside, near the
                                        boundary, outside) and maps.interp3d (linear / cubic) in/out pairs
  logs.npz                              the reference's verb=4 log text (cycle-QC figure) of small V / W / F solves
  receivers_modes.npz                   maps.interp3d cubic with mode 'nearest' / 'mirror' / 'reflect' / 'wrap', fields.get_receiver(extrapolate=True)
  solves_div.npz                        the DIVERGED case of the reference's test_solver_heterogeneous (2**9 x 2 x 2)
  solves_entry.npz                      small odd grids where the first sc_dir has clevel 0 (level 0's cycmax is
                                        fixed on entry of solver.multigrid)
  (--big) solves_32.npz                 32^3 config-C1 plumbing case
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    tmp = tempfile.mkdtemp(prefix="emg3d_ref_stub_")
    os.makedirs(os.path.join(tmp, "numba"))
    with open(os.path.join(tmp, "numba", "__init__.py"), "w") as f:
        f.write(
            "__version__ = '0.0-stub'\n"
            "def njit(*a, **k):\n"
            "    def deco(f):\n"
            "        f.py_func = f\n"
            "        return f\n"
            "    return deco(a[0]) if (len(a) == 1 and callable(a[0]) and not k) else deco\n"
            "jit = njit\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    np.infty = np.inf
    np.float_ = np.float64
    np.complex_ = np.complex128
    import scipy.sparse.linalg as ssl
    for name in ("bicgstab", "cgs", "gcrotmk"):
        orig = getattr(ssl, name)

        def wrapped(*a, _orig=orig, **k):
            if "tol" in k:
                k["rtol"] = k.pop("tol")
            return _orig(*a, **k)
        setattr(ssl, name, wrapped)
    # SciPy >= 1.14 moved a private helper the reference calls (maps.py:243)
    import scipy.interpolate as _si
    if not hasattr(_si.interpnd, '_ndim_coords_from_arrays'):
        try:
            _si.interpnd._ndim_coords_from_arrays
        except AttributeError:
            import types
            from scipy.interpolate import _interpnd
            _si.interpnd = types.SimpleNamespace(_ndim_coords_from_arrays=_interpnd._ndim_coords_from_arrays)
    import emg3d  # noqa
    return emg3d


def get_h(ncore, npad, width, factor):
    """Stretched widths (same formula as the reference's test helper
    tests/test_meshes.py:28-31; data generator, not product code)."""
    pad = ((np.ones(npad) * np.abs(factor)) ** (np.arange(npad) + 1)) * width
    return np.r_[pad[::-1], np.ones(ncore) * width, pad]


def kernel_fixture(emg3d, dtype, seed):
    from emg3d import core, solver, fields, meshes, models
    rng = np.random.default_rng(seed)
    hx = rng.uniform(20, 60, 8)
    hy = rng.uniform(20, 60, 6)
    hz = rng.uniform(20, 60, 4)
    origin = np.array([-100., 50., -30.])
    grid = meshes.TensorMesh([hx, hy, hz], origin=origin)
    freq = 1.3 if dtype == np.complex128 else -1.3
    rho = 10 ** rng.uniform(-0.5, 1.5, (3, grid.nC))
    mu_r = rng.uniform(0.8, 1.5, grid.nC)
    model = models.Model(grid, rho[0], rho[1], rho[2], mu_r=mu_r)
    sf = fields.SourceField(grid, freq=freq)
    vm = models.VolumeModel(grid, model, sf)

    def rfield(pec=True):
        v = rng.standard_normal(grid.nE)
        if dtype == np.complex128:
            v = v + 1j * rng.standard_normal(grid.nE)
        f = fields.Field(grid, v.astype(dtype), freq=freq)
        if pec:
            f.ensure_pec
        return f

    out = {'hx': hx, 'hy': hy, 'hz': hz, 'origin': origin, 'freq': freq,
           'eta_x': vm.eta_x, 'eta_y': vm.eta_y, 'eta_z': vm.eta_z, 'zeta': vm.zeta,
           'smu0': np.array(sf.smu0)}

    e = rfield()
    s = rfield() * 1e-3
    out['e'] = np.array(e)
    out['s'] = np.array(s)

    # amat_x / residual
    r = s.copy()
    core.amat_x(r.fx, r.fy, r.fz, e.fx, e.fy, e.fz, vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta,
                hx, hy, hz)
    out['amat_x_r'] = np.array(r)

    # smoothers
    for name, fn in (('gs', core.gauss_seidel), ('gs_x', core.gauss_seidel_x),
                     ('gs_y', core.gauss_seidel_y), ('gs_z', core.gauss_seidel_z)):
        for nu in (1, 2, 3):
            ee = e.copy()
            fn(ee.fx, ee.fy, ee.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta,
               hx, hy, hz, nu)
            out[f'{name}_nu{nu}'] = np.array(ee)

    # smoothing() dispatch for every lr_dir (solver.py:738-799)
    for lr_dir in range(8):
        ee = e.copy()
        solver.smoothing(grid, vm, s, ee, 2, lr_dir)
        out[f'smoothing_lr{lr_dir}'] = np.array(ee)

    # restriction (grid, model, field) + prolongation for every sc_dir
    res = rfield(pec=False)
    out['res'] = np.array(res)
    for sc_dir in range(7):
        cgrid, cmodel, csfield, cefield = solver.restriction(grid, vm, s, res, sc_dir)
        out[f'restrict{sc_dir}_chx'] = cgrid.h[0]
        out[f'restrict{sc_dir}_chy'] = cgrid.h[1]
        out[f'restrict{sc_dir}_chz'] = cgrid.h[2]
        out[f'restrict{sc_dir}_eta_x'] = cmodel.eta_x
        out[f'restrict{sc_dir}_eta_y'] = cmodel.eta_y
        out[f'restrict{sc_dir}_eta_z'] = cmodel.eta_z
        out[f'restrict{sc_dir}_zeta'] = cmodel.zeta
        out[f'restrict{sc_dir}_csfield'] = np.array(csfield)
        wx, wy, wz = solver._get_restriction_weights(grid, cgrid, sc_dir)
        for nm, w in (('wx', wx), ('wy', wy), ('wz', wz)):
            for q, a in zip('l0r', w):
                out[f'restrict{sc_dir}_{nm}{q}'] = a
        v = rng.standard_normal(cgrid.nE)
        if dtype == np.complex128:
            v = v + 1j * rng.standard_normal(cgrid.nE)
        ce = fields.Field(cgrid, v.astype(dtype), freq=freq)
        out[f'prolong{sc_dir}_ce'] = np.array(ce)
        ee = e.copy()
        solver.prolongation(grid, ee, cgrid, ce, sc_dir)
        out[f'prolong{sc_dir}_e'] = np.array(ee)

    # residual norm
    out['residual_norm'] = solver.residual(grid, vm, s, e, True)
    return out


def regression_fixture(emg3d):
    """Re-export inputs + goldens of the reference's tests/data/regression.npz
    and capture per-cycle traces by running the reference here."""
    from emg3d import solver, fields, meshes, models
    dat = np.load(os.path.join(REF, "tests", "data", "regression.npz"), allow_pickle=True)
    out = {}

    def mk(prefix, inp=None):
        hx, hy, hz = (dat[f'{prefix}>grid>h{c}'] for c in 'xyz')
        origin = dat[f'{prefix}>grid>origin']
        grid = meshes.TensorMesh([hx, hy, hz], origin=origin)
        px, py, pz = (dat[f'{prefix}>model>property_{c}'] for c in 'xyz')
        model = models.Model(grid, px, py, pz)
        freq = float(dat[f'{prefix}>sfield>freq'])
        sfield = fields.SourceField(grid, dat[f'{prefix}>sfield>field'], freq=freq)
        for k, v in (('hx', hx), ('hy', hy), ('hz', hz), ('origin', origin), ('property_x', px),
                     ('property_y', py), ('property_z', pz), ('freq', freq),
                     ('sfield', dat[f'{prefix}>sfield>field'])):
            out[f'{prefix}_{k}'] = v
        vm = models.VolumeModel(grid, model, sfield)
        out[f'{prefix}_smu0_here'] = np.array(sfield.smu0)
        out[f'{prefix}_eta_x_here'] = vm.eta_x
        return grid, model, sfield

    # res: F, W, V, bicgstab
    grid, model, sfield = mk('res')
    out['res_src'] = dat['res>input_source>src']
    for key, kw in (('F', {}), ('W', {'cycle': 'W'}), ('V', {'cycle': 'V'}),
                    ('bic', {'sslsolver': True})):
        out[f'res_{key}_golden'] = dat[f'res>{key}result>field']
        ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, **kw)
        out[f'res_{key}_here'] = np.array(ef)
        out[f'res_{key}_error_at_cycle'] = info['error_at_cycle']
        out[f'res_{key}_it'] = np.array([info['it_mg'], info['it_ssl']])
        print('res', key, info['it_mg'], info['it_ssl'], info['rel_error'],
              np.abs(ef - dat[f'res>{key}result>field']).max() / np.abs(ef).max())

    # reg_2
    grid, model, sfield = mk('reg_2')
    inp = {k: dat[f'reg_2>inp>{k}'].item() for k in
           ('semicoarsening', 'linerelaxation', 'tol', 'maxit', 'nu_init', 'nu_pre',
            'nu_coarse', 'nu_post', 'clevel')}
    for k, v in inp.items():
        out[f'reg_2_inp_{k}'] = v
    out['reg_2_golden'] = dat['reg_2>result>field']
    ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, **inp)
    out['reg_2_here'] = np.array(ef)
    out['reg_2_error_at_cycle'] = info['error_at_cycle']
    print('reg_2', info['it_mg'], info['rel_error'],
          np.abs(ef - dat['reg_2>result>field']).max() / np.abs(ef).max())

    # lap (float64)
    grid, model, sfield = mk('lap')
    out['lap_src'] = dat['lap>input_source>src']
    for key, kw in (('F', {}), ('bic', {'sslsolver': True})):
        out[f'lap_{key}_golden'] = dat[f'lap>{key}result>field']
        ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, **kw)
        out[f'lap_{key}_here'] = np.array(ef)
        out[f'lap_{key}_error_at_cycle'] = info['error_at_cycle']
        out[f'lap_{key}_it'] = np.array([info['it_mg'], info['it_ssl']])
        print('lap', key, info['it_mg'], info['it_ssl'], info['rel_error'])
    return out


def solves_fixture(emg3d, n, ncore, npad, kinds):
    from emg3d import solver, fields, meshes, models
    h = get_h(ncore, npad, 100, 1.3)
    hz = get_h(ncore, npad, 100, 1.35)
    origin = np.array([-h.sum() / 2, -h.sum() / 2, -hz.sum() / 2])
    grid = meshes.TensorMesh([h, h, hz], origin=origin)
    rng = np.random.default_rng(1234)
    rho_b = 10 ** rng.uniform(-0.5, 1.5, grid.nC)
    model = models.Model(grid, rho_b, 2 * rho_b, 3 * rho_b)
    src = [0., 0., 0., 30., 10.]
    sfield = fields.get_source_field(grid, src, 1.0)
    vm = models.VolumeModel(grid, model, sfield)
    out = {'hx': h, 'hy': h, 'hz': hz, 'origin': origin, 'rho_b': rho_b, 'src': np.array(src),
           'freq': 1.0, 'sfield': np.array(sfield), 'smu0': np.array(sfield.smu0),
           'eta_x': vm.eta_x, 'zeta': vm.zeta}
    for name, kw in kinds.items():
        ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, **kw)
        out[f'{name}_efield'] = np.array(ef)
        out[f'{name}_error_at_cycle'] = info['error_at_cycle']
        out[f'{name}_it'] = np.array([info['it_mg'], info['it_ssl']])
        out[f'{name}_exit'] = np.array(info['exit'])
        print(n, name, info['it_mg'], info['it_ssl'], info['rel_error'], info['exit_message'])
    return out


def source_fixture(emg3d):
    from emg3d import fields, meshes
    out = {}
    rng = np.random.default_rng(7)
    hx = rng.uniform(20, 60, 8); hy = rng.uniform(20, 60, 6); hz = rng.uniform(20, 60, 10)
    origin = np.array([-150., -100., -200.])
    grid = meshes.TensorMesh([hx, hy, hz], origin=origin)
    out.update(hx=hx, hy=hy, hz=hz, origin=origin)
    cases = {
        'point': ([10., -5., 20., 30., 10.], 1.0),
        'point_lap': ([-33., 21., -50., 120., -40.], -2.5),
        'dipole': ([-40., 55., -20., 31., -80., 10.], 0.5),
        'dipole_x': ([-40., 55., 0., 0., 10., 10.], 2.0),
    }
    for k, (src, freq) in cases.items():
        sf = fields.get_source_field(grid, src, freq)
        out[f'{k}_src'] = np.array(src)
        out[f'{k}_freq'] = freq
        out[f'{k}_sfield'] = np.array(sf)
        out[f'{k}_smu0'] = np.array(sf.smu0)
    # arbitrarily shaped and magnetic sources (fields.py:538-577, 1043-1049): (src, freq, strength, electric, length)
    path = [[-60., -10., 35., 70.], [-40., -35., 10., 45.], [-90., -30., -25., 40.]]
    more = {
        'shaped': (path, 1.0, 0, True, 1.0),
        'shaped_strength': (path, 2.0, 3 - 2j, True, 1.0),
        'loop': ([10., -5., 20., 30., 10.], 1.0, 0, False, 1.0),            # magnetic point dipole = square loop
        'loop_big': ([-20., 15., -40., 120., -35.], -1.5, 2.5, False, 30.),
        'shaped_mag': (path, 0.5, 0, False, 1.0),
        'point_len': ([10., -5., 20., 30., 10.], 1.0, 4.0, True, 25.),
    }
    # long oblique dipoles: the reference's cell loop also visits cells of the bounding box that the dipole does not cross
    # and normalises the result afterwards ("Normalizing Source", fields.py:1003-1010); reversed directions included
    nx_, ny_, nz_ = grid.nodes_x, grid.nodes_y, grid.nodes_z
    r2 = np.random.default_rng(77)
    for i in range(6):
        ends = [r2.uniform(n_[0] + 1, n_[-1] - 1, 2) for n_ in (nx_, ny_, nz_)]
        more[f'oblique_{i}'] = ([ends[0][0], ends[0][1], ends[1][0], ends[1][1], ends[2][0], ends[2][1]],
                                [1.0, -1.0][i % 2], 0, True, 1.0)
    for k, (src, freq, strength, electric, length) in more.items():
        sf = fields.get_source_field(grid, src, freq, strength=strength, electric=electric, length=length)
        out[f'{k}_src'] = np.array(src)
        out[f'{k}_freq'] = freq
        out[f'{k}_strength'] = np.array(strength)
        out[f'{k}_electric'] = electric
        out[f'{k}_length'] = length
        out[f'{k}_sfield'] = np.array(sf)
        out[f'{k}_moment'] = np.array(sf.moment)
    return out


def receivers_fixture(emg3d):
    """fields.get_receiver_response (fields.py:733-817) and maps.interp3d (maps.py:179-276)."""
    from emg3d import fields, meshes, models, maps
    out = {}
    rng = np.random.default_rng(21)
    hx = rng.uniform(20, 60, 12); hy = rng.uniform(20, 60, 9); hz = rng.uniform(20, 60, 10)
    origin = np.array([-250., -180., -220.])
    grid = meshes.TensorMesh([hx, hy, hz], origin=origin)
    out.update(hx=hx, hy=hy, hz=hz, origin=origin)
    # a smooth complex field + noise (so that the spline interpolation has structure to follow)
    ef = fields.Field(grid, freq=1.3)
    for c, f in enumerate((ef.fx, ef.fy, ef.fz)):
        ax = [np.linspace(0, 1, n) for n in f.shape]
        X, Y, Z = np.meshgrid(*ax, indexing='ij')
        f[...] = (np.sin(3 * X + c) * np.cos(2 * Y) * (1 + Z) + 1j * np.cos(2 * X - Z + c) * np.sin(3 * Y)
                  + 0.1 * (rng.standard_normal(f.shape) + 1j * rng.standard_normal(f.shape)))
    out['efield'] = np.array(ef)
    out['freq'] = 1.3
    nrec = 14
    rx = rng.uniform(grid.nodes_x[1], grid.nodes_x[-2], nrec)
    ry = rng.uniform(grid.nodes_y[1], grid.nodes_y[-2], nrec)
    rz = rng.uniform(grid.nodes_z[1], grid.nodes_z[-2], nrec)
    # on a node, in the first / last interior interval, outside the trimmed grid, outside the grid
    rx[0], ry[0], rz[0] = grid.nodes_x[3], grid.nodes_y[4], grid.nodes_z[5]
    rx[1], ry[1], rz[1] = grid.nodes_x[1] + 1.0, grid.nodes_y[1] + 2.0, grid.nodes_z[1] + 0.5
    rx[2], ry[2], rz[2] = grid.nodes_x[-2] - 1.0, grid.nodes_y[-2] - 2.0, grid.nodes_z[-2] - 0.5
    rx[3] = grid.nodes_x[0] + 1.0
    rx[4] = grid.nodes_x[-1] + 50.0
    azm = rng.uniform(-180, 180, nrec); dip = rng.uniform(-90, 90, nrec)
    azm[5], dip[5] = 0., 0.
    azm[6], dip[6] = 90., 0.
    azm[7], dip[7] = 0., 90.
    rec = (rx, ry, rz, azm, dip)
    out['rec'] = np.stack(rec)
    out['resp_e'] = np.array(fields.get_receiver_response(grid, ef, rec))
    # scalar angles (broadcast), one component switched off by the 1e-10 rule
    out['resp_e_x'] = np.array(fields.get_receiver_response(grid, ef, (rx, ry, rz, 0., 0.)))
    out['resp_e_z'] = np.array(fields.get_receiver_response(grid, ef, (rx, ry, rz, 20., 90.)))
    # magnetic field
    rho = 10 ** rng.uniform(-0.5, 1.5, grid.nC)
    model = models.Model(grid, rho, 2 * rho, 3 * rho)
    hf = fields.get_h_field(grid, model, ef)
    out['rho'] = rho
    out['hfield'] = np.array(hf)
    out['resp_h'] = np.array(fields.get_receiver_response(grid, hf, rec))
    # Laplace-domain (real) field
    lf = fields.Field(grid, np.array(ef).real.copy(), freq=-2.0)
    out['resp_lap'] = np.array(fields.get_receiver_response(grid, lf, rec))
    # maps.interp3d directly: linear and cubic, fill / cval variants
    pts = (grid.cell_centers_x, grid.nodes_y, grid.nodes_z)
    vals = np.asfortranarray(ef.fx)
    xi = (rx, ry, rz)
    out['i3d_linear_fill0'] = maps.interp3d(pts, vals, xi, 'linear', 0.0, 'constant', 0.0)
    out['i3d_linear_extrap'] = maps.interp3d(pts, vals, xi, 'linear', None, 'constant', 0.0)
    out['i3d_cubic_nan'] = maps.interp3d(pts, vals, xi, 'cubic', 0.0, 'constant', np.nan)
    out['i3d_cubic_c0'] = maps.interp3d(pts, vals, xi, 'cubic', 0.0, 'constant', 0.0)
    # fields.get_receiver (fields.py:634-730): whole field (tuple), one component, a model parameter; with / without
    # extrapolation (linear only: the cubic + extrapolate combination uses map_coordinates' mode='nearest')
    gx, gy, gz = fields.get_receiver(grid, ef, xi)
    out['getrec_cubic_fx'], out['getrec_cubic_fy'], out['getrec_cubic_fz'] = np.array(gx), np.array(gy), np.array(gz)
    out['getrec_linear_fy'] = np.array(fields.get_receiver(grid, ef.fy, xi, 'linear'))
    out['getrec_linear_extrap_fz'] = np.array(fields.get_receiver(grid, ef.fz, xi, 'linear', True))
    out['getrec_hx'] = np.array(fields.get_receiver(grid, hf.fx, xi))
    out['getrec_rho_linear'] = np.array(fields.get_receiver(grid, rho.reshape(grid.vnC, order='F'), xi, 'linear'))
    out['getrec_rho_cubic'] = np.array(fields.get_receiver(grid, rho.reshape(grid.vnC, order='F'), xi, 'cubic'))
    # a grid with 3-cell axes: the trimmed cell-centre axis has ONE point (RegularGridInterpolator: fill value unless the
    # coordinate equals that point; forced linear), the trimmed node axis two
    r3 = np.random.default_rng(5)
    for tag, shape in (('s635', (6, 3, 5)), ('s333', (3, 3, 3)), ('s734', (7, 3, 4))):
        hs = [r3.uniform(10, 80, n) for n in shape]
        og = np.array([-40., 5., -100.])
        gs = meshes.TensorMesh(hs, origin=og)
        fs = fields.Field(gs, r3.standard_normal(gs.nE) + 1j * r3.standard_normal(gs.nE), freq=1.0)
        n_ = 9
        rxs = r3.uniform(gs.nodes_x[0], gs.nodes_x[-1], n_); rys = r3.uniform(gs.nodes_y[0], gs.nodes_y[-1], n_)
        rzs = r3.uniform(gs.nodes_z[0], gs.nodes_z[-1], n_)
        rys[0] = gs.cell_centers_y[1]; rxs[1] = gs.nodes_x[1]; rzs[2] = gs.nodes_z[2]      # exactly on the single points
        rys[3] = gs.cell_centers_y[1]; rxs[3] = gs.cell_centers_x[1]; rzs[3] = gs.cell_centers_z[1]
        recs = (rxs, rys, rzs, r3.uniform(-180, 180, n_), r3.uniform(-90, 90, n_))
        out[f'{tag}_hx'], out[f'{tag}_hy'], out[f'{tag}_hz'], out[f'{tag}_origin'] = hs[0], hs[1], hs[2], og
        out[f'{tag}_field'] = np.array(fs)
        out[f'{tag}_rec'] = np.stack(recs)
        out[f'{tag}_resp'] = np.array(fields.get_receiver_response(gs, fs, recs))
    # receivers exactly ON the first / last point of the trimmed grid (and on interior nodes) of short axes (4 ... 8
    # trimmed points: cubic): the reference returns values there, not NaN
    r4 = np.random.default_rng(9)
    for tag, shape in (('e557', (5, 5, 7)), ('e659', (6, 5, 9))):
        hs = [r4.uniform(10, 80, n) for n in shape]
        og = r4.uniform(-500, 100, 3)
        gs = meshes.TensorMesh(hs, origin=og)
        fs = fields.Field(gs, r4.standard_normal(gs.nE) + 1j * r4.standard_normal(gs.nE), freq=1.0)
        n_ = 40
        cs = []
        for nd in (gs.nodes_x, gs.nodes_y, gs.nodes_z):
            c = r4.uniform(nd[1], nd[-2], n_)
            k = r4.integers(0, n_, 24)
            c[k] = r4.choice([nd[1], nd[-2], nd[2], nd[-3]], k.size)
            cs.append(c)
        recs = (cs[0], cs[1], cs[2], r4.uniform(-180, 180, n_), r4.uniform(-90, 90, n_))
        out[f'{tag}_hx'], out[f'{tag}_hy'], out[f'{tag}_hz'], out[f'{tag}_origin'] = hs[0], hs[1], hs[2], og
        out[f'{tag}_field'] = np.array(fs)
        out[f'{tag}_rec'] = np.stack(recs)
        out[f'{tag}_resp'] = np.array(fields.get_receiver_response(gs, fs, recs))
    return out


def receiver_modes_fixture(emg3d):
    """maps.interp3d with the cubic boundary modes 'nearest', 'mirror', 'reflect' and 'wrap' (maps.py:249-272: scipy.ndimage.map_coordinates)
    and fields.get_receiver(extrapolate=True) with the cubic method (fields.py:717-724: mode='nearest'), on the grid and
    field of receivers.npz; coordinates inside, on the boundary, up to 1.5 cells and far outside the trimmed grid."""
    from emg3d import fields, meshes, maps
    g = np.load(os.path.join(HERE, 'receivers.npz'))
    grid = meshes.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    ef = fields.Field(grid, g['efield'].copy(), freq=float(g['freq']))
    rng = np.random.default_rng(77)
    n = 40
    lo = [grid.nodes_x[0] - 60., grid.nodes_y[0] - 60., grid.nodes_z[0] - 60.]
    hi = [grid.nodes_x[-1] + 60., grid.nodes_y[-1] + 60., grid.nodes_z[-1] + 60.]
    xi = [rng.uniform(lo[i], hi[i], n) for i in range(3)]
    xi[0][0], xi[1][0], xi[2][0] = grid.nodes_x[1], grid.nodes_y[1], grid.nodes_z[1]            # first trimmed node
    xi[0][1], xi[1][1], xi[2][1] = grid.nodes_x[-2], grid.nodes_y[-2], grid.nodes_z[-2]         # last trimmed node
    xi[0][2] = grid.nodes_x[-1] + 900.                                                          # far outside
    xi[1][3] = grid.nodes_y[0] - 700.
    out = {'xi': np.stack(xi)}
    pts = (grid.cell_centers_x, grid.nodes_y, grid.nodes_z)
    vals = np.asfortranarray(ef.fx)
    for mode in ('nearest', 'mirror', 'reflect', 'wrap'):
        out[f'i3d_cubic_{mode}'] = maps.interp3d(pts, vals, tuple(xi), 'cubic', 0.0, mode, 0.0)
        out[f'i3d_cubic_{mode}_real'] = maps.interp3d(pts, vals.real.copy(), tuple(xi), 'cubic', 0.0, mode, 0.0)
    gx, gy, gz = fields.get_receiver(grid, ef, tuple(xi), 'cubic', True)
    out['getrec_cubic_extrap_fx'], out['getrec_cubic_extrap_fy'], out['getrec_cubic_extrap_fz'] = (
        np.array(gx), np.array(gy), np.array(gz))
    return out


def logs_fixture(emg3d):
    """The reference's log text at verb=4 (solver.py:1575-1648: cycle-QC figure in front of the first cycle's line) for
    V / W / F cycles with and without semicoarsening on small grids; two cycles each."""
    from emg3d import solver, fields, meshes, models
    out = {}
    cases = [('F16', (16, 16, 16), dict(cycle='F')), ('V16sc', (16, 16, 16), dict(cycle='V', semicoarsening=True, linerelaxation=True)),
             ('W16', (16, 16, 16), dict(cycle='W')), ('F32x8sc', (32, 8, 8), dict(cycle='F', semicoarsening=2)),
             ('F24', (24, 12, 6), dict(cycle='F', semicoarsening=True)), ('Fclev1', (16, 16, 16), dict(cycle='F', clevel=1)),
             ('W64', (64, 4, 4), dict(cycle='W')), ('F2', (2, 2, 2), dict(cycle='F'))]
    for tag, shape, kw in cases:
        h = [np.ones(n) * 50. for n in shape]
        grid = meshes.TensorMesh(h, origin=[-hh.sum() / 2 for hh in h])
        model = models.Model(grid, 1.5)
        sfield = fields.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0)
        _, info = solver.solve(grid, model, sfield, verb=4, log=-1, maxit=2, tol=1e-30, return_info=True, **kw)
        out[f'{tag}_log'] = np.array(info['log'])
        out[f'{tag}_shape'] = np.array(shape)
        out[f'{tag}_kw'] = np.array(repr(kw))
    out['cases'] = np.array([c[0] for c in cases])
    # verb=5 (solver.py:498-578): the norm after every smoothing call of every level; also with initial smoothing, without
    # pre-smoothing, and as a BiCGSTAB preconditioner
    cases5 = [('v5_F16', (16, 16, 16), dict(cycle='F')), ('v5_V16sc', (16, 16, 16), dict(cycle='V', semicoarsening=True, linerelaxation=True)),
              ('v5_W8', (8, 8, 16), dict(cycle='W', nu_init=2)), ('v5_F24', (24, 12, 6), dict(cycle='F', semicoarsening=True, nu_pre=0)),
              ('v5_F2', (2, 2, 2), dict(cycle='F')), ('v5_bicg', (8, 8, 8), dict(cycle='F', sslsolver='bicgstab'))]
    for tag, shape, kw in cases5:
        h = [np.ones(n) * 50. for n in shape]
        grid = meshes.TensorMesh(h, origin=[-hh.sum() / 2 for hh in h])
        model = models.Model(grid, 1.5)
        sfield = fields.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0)
        _, info = solver.solve(grid, model, sfield, verb=5, log=-1, maxit=2, tol=1e-30, return_info=True, **kw)
        out[f'{tag}_log'] = np.array(info['log'])
        out[f'{tag}_shape'] = np.array(shape)
        out[f'{tag}_kw'] = np.array(repr(kw))
    out['cases5'] = np.array([c[0] for c in cases5])
    return out


def gradient_fixture(emg3d):
    """What optimize.gradient (optimize.py:115-217) and Simulation._get_rfield / _get_bfields
    (simulations.py:1131-1213) compute for ONE (source, frequency) pair, composed from the reference's own
    functions (Simulation / Survey need xarray, which is not installed here)."""
    from emg3d import fields, meshes, models, maps, solver
    out = {}
    rng = np.random.default_rng(31)
    hx = get_h(6, 3, 80., 1.3); hy = get_h(4, 3, 80., 1.3); hz = get_h(4, 2, 80., 1.4)
    origin = np.array([-hx.sum() / 2, -hy.sum() / 2, -hz.sum() / 2])
    grid = meshes.TensorMesh([hx, hy, hz], origin=origin)
    out.update(hx=hx, hy=hy, hz=hz, origin=origin)
    # edges2cellaverages alone, on random complex and real "fields"
    f = fields.Field(grid, rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE), freq=1.)
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    for tag, ff in (('c', f), ('r', fields.Field(grid, np.array(f).real.copy(), freq=-1.))):
        ox = np.zeros(grid.vnC, order='F', dtype=ff.dtype); oy = ox.copy(); oz = ox.copy()
        maps.edges2cellaverages(ex=ff.fx, ey=ff.fy, ez=ff.fz, vol=vol, out_x=ox, out_y=oy, out_z=oz)
        out[f'e2c_{tag}_in'] = np.array(ff)
        out[f'e2c_{tag}_x'], out[f'e2c_{tag}_y'], out[f'e2c_{tag}_z'] = ox, oy, oz
    # the adjoint-state gradient of one source / frequency (isotropic model: optimize.py:161-163)
    freq = 1.5
    res = 10 ** rng.uniform(-0.3, 1.0, grid.nC)
    model = models.Model(grid, res)
    res_true = res.copy().reshape(grid.vnC, order='F')
    res_true[5:9, 3:7, 2:5] *= 4.0                       # the "observed" data come from a perturbed model
    src = np.array([-100., 30., 20., 25., 5.])
    nrec = 5
    rec = (np.array([150., 220., -260., 60., 300.]), np.array([40., -90., 110., 10., -30.]),
           np.array([-20., 30., 10., -50., 25.]), np.array([0., 40., -70., 90., 10.]), np.array([0., 10., -15., 30., 60.]))
    opts = dict(cycle='F', semicoarsening=True, linerelaxation=True, tol=1e-8, verb=1)
    sfield = fields.get_source_field(grid, src, freq)
    efield = solver.solve(grid, model, sfield, **opts)
    e_obs = solver.solve(grid, models.Model(grid, res_true.ravel('F')), sfield, **opts)
    syn = np.array(fields.get_receiver_response(grid, efield, rec))
    obs = np.array(fields.get_receiver_response(grid, e_obs, rec))
    weights = 1.0 / (0.05 * np.abs(obs)) ** 2            # relative error 5 % (data weights = 1 / std^2)
    residual = syn - obs
    misfit = np.sum(weights * (residual.conj() * residual)).real / 2           # optimize.py:110
    rfield = fields.SourceField(grid, freq=freq)                                # simulations.py:1171-1213
    for i in range(nrec):
        strength = residual[i].conj() * np.conj(weights[i]) / rfield.smu0
        rfield += fields.get_source_field(grid=grid, src=[r[i] for r in rec], freq=freq, strength=strength)
    bfield = solver.solve(grid, model, rfield, **opts)                          # simulations.py:1131-1143
    prod = -np.real(bfield * efield * efield.smu0)                              # optimize.py:181-184
    prod = fields.Field(grid, prod.astype(np.float64), freq=-1.)
    gx = np.zeros(grid.vnC, order='F'); gy = gx.copy(); gz = gx.copy()
    maps.edges2cellaverages(ex=prod.fx, ey=prod.fy, ez=prod.fz, vol=vol, out_x=gx, out_y=gy, out_z=gz)
    out.update(freq=freq, res=res, src=src, rec=np.stack(rec), observed=obs, weights=weights, synthetic=syn,
               misfit=misfit, rfield=np.array(rfield), efield=np.array(efield), bfield=np.array(bfield),
               smu0=np.array(efield.smu0), grad=gx + gy + gz)
    # the same with MAGNETIC receivers: data from H = get_h_field(E) (simulations.py: _get_responses), residual sources
    # are magnetic point dipoles (square loops) of strength conj(r) conj(w) / smu0 / smu0 (simulations.py:1190-1197)
    hf = fields.get_h_field(grid, model, efield)
    hf_obs = fields.get_h_field(grid, models.Model(grid, res_true.ravel('F')), e_obs)
    msyn = np.array(fields.get_receiver_response(grid, hf, rec))
    mobs = np.array(fields.get_receiver_response(grid, hf_obs, rec))
    mweights = 1.0 / (0.05 * np.abs(mobs)) ** 2
    mres = msyn - mobs
    mmisfit = np.sum(mweights * (mres.conj() * mres)).real / 2
    mrfield = fields.SourceField(grid, freq=freq)
    for i in range(nrec):
        strength = mres[i].conj() * np.conj(mweights[i]) / mrfield.smu0 / mrfield.smu0
        mrfield += fields.get_source_field(grid=grid, src=[r[i] for r in rec], freq=freq, strength=strength,
                                           electric=False)
    mbfield = solver.solve(grid, model, mrfield, **opts)
    mprod = fields.Field(grid, (-np.real(mbfield * efield * efield.smu0)).astype(np.float64), freq=-1.)
    gx = np.zeros(grid.vnC, order='F'); gy = gx.copy(); gz = gx.copy()
    maps.edges2cellaverages(ex=mprod.fx, ey=mprod.fy, ez=mprod.fz, vol=vol, out_x=gx, out_y=gy, out_z=gz)
    out.update(m_observed=mobs, m_weights=mweights, m_synthetic=msyn, m_misfit=mmisfit, m_rfield=np.array(mrfield),
               m_bfield=np.array(mbfield), m_grad=gx + gy + gz)
    return out


def solves_eps_fixture(emg3d):
    """16^3 stretched tri-axial model WITH epsilon_r (and mu_r): eta = s mu_0 V (sigma - s eps_0 eps_r)
    (emg3d/models.py:631-647) in the frequency and in the Laplace domain -- the arrays themselves and an F-cycle solve each.
    eps_r is chosen large enough (3e4 ... 3e6 at 20 Hz; not physical) that the displacement term is 10-30 % of sigma."""
    from emg3d import solver, fields, meshes, models
    h = get_h(8, 4, 100, 1.3)
    hz = get_h(8, 4, 100, 1.35)
    origin = np.array([-h.sum() / 2, -h.sum() / 2, -hz.sum() / 2])
    grid = meshes.TensorMesh([h, h, hz], origin=origin)
    rng = np.random.default_rng(4321)
    rho_b = 10 ** rng.uniform(-0.5, 1.5, grid.nC)
    eps_r = 10 ** rng.uniform(4.5, 6.5, grid.nC)
    mu_r = rng.uniform(1., 2., grid.nC)
    src = [0., 0., 0., 30., 10.]
    out = {'hx': h, 'hy': h, 'hz': hz, 'origin': origin, 'rho_b': rho_b, 'eps_r': eps_r, 'mu_r': mu_r, 'src': np.array(src)}
    for tag, freq in (("f", 20.0), ("s", -20.0)):
        model = models.Model(grid, rho_b, 2 * rho_b, 3 * rho_b, mu_r=mu_r, epsilon_r=eps_r)
        sfield = fields.get_source_field(grid, src, freq)
        vm = models.VolumeModel(grid, model, sfield)
        ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, cycle='F', semicoarsening=True,
                                linerelaxation=True)
        out.update({f'{tag}_freq': freq, f'{tag}_eta_x': vm.eta_x, f'{tag}_eta_y': vm.eta_y, f'{tag}_eta_z': vm.eta_z,
                    f'{tag}_zeta': vm.zeta, f'{tag}_efield': np.array(ef), f'{tag}_error_at_cycle': info['error_at_cycle'],
                    f'{tag}_it': np.array(info['it_mg']), f'{tag}_exit': np.array(info['exit'])})
        print('eps', tag, info['it_mg'], info['rel_error'], info['exit_message'])
    return out


def colour_replay(core, nC, e, s, eta, zeta, h, direction, nu, lex=False):
    F = np.asfortranarray
    fwd, bwd = (1, 3, 0, 2), (0, 3, 2, 1)
    """e: flat field, updated in place through its three views."""
    nx, ny, nz = nC
    nEx, nEy = nx * (ny + 1) * (nz + 1), (nx + 1) * ny * (nz + 1)
    ex = e[:nEx].reshape((nx, ny + 1, nz + 1), order='F')
    ey = e[nEx:nEx + nEy].reshape((nx + 1, ny, nz + 1), order='F')
    ez = e[nEx + nEy:].reshape((nx + 1, ny + 1, nz), order='F')
    sx = s[:nEx].reshape(ex.shape, order='F')
    sy = s[nEx:nEx + nEy].reshape(ey.shape, order='F')
    sz = s[nEx + nEy:].reshape(ez.shape, order='F')
    full = slice(None)

    def one(ix, iy, iz):
        # node index None = the line direction: the whole axis
        def nodes(i):
            return full if i is None else slice(i - 1, i + 2)

        def cells(i):
            return full if i is None else slice(i - 1, i + 1)
        xs, ys, zs = nodes(ix), nodes(iy), nodes(iz)
        xc, yc, zc = cells(ix), cells(iy), cells(iz)
        a = [F(ex[xc, ys, zs]), F(ey[xs, yc, zs]), F(ez[xs, ys, zc])]
        fn = [core.gauss_seidel, core.gauss_seidel_x, core.gauss_seidel_y, core.gauss_seidel_z][direction]
        fn(a[0], a[1], a[2], F(sx[xc, ys, zs]), F(sy[xs, yc, zs]), F(sz[xs, ys, zc]),
           F(eta[0][xc, yc, zc]), F(eta[1][xc, yc, zc]), F(eta[2][xc, yc, zc]), F(zeta[xc, yc, zc]),
           h[0][xc], h[1][yc], h[2][zc], 1)
        ex[xc, ys, zs], ey[xs, yc, zs], ez[xs, ys, zc] = a

    if lex:
        # the reference's own first sweep (descending lexicographic), node by node / line by line: must reproduce the
        # full-grid call bit for bit (checked below against the kernel fixtures) -- the proof that `one` is the
        # reference's single update
        rng_ = [range(n - 1, 0, -1) for n in nC]
        if direction == 0:
            for iz in rng_[2]:
                for iy in rng_[1]:
                    for ix in rng_[0]:
                        one(ix, iy, iz)
        else:
            P, Q = {1: (1, 2), 2: (0, 2), 3: (0, 1)}[direction]
            for iQ in rng_[Q]:
                for iP in rng_[P]:
                    idx = [None, None, None]
                    idx[P], idx[Q] = iP, iQ
                    one(*idx)
        return
    iback = 0
    for _ in range(nu):
        iback = 1 - iback
        if direction == 0:
            for col in range(8):
                for iz in range(1 + ((col >> 2) & 1), nz, 2):
                    for iy in range(1 + ((col >> 1) & 1), ny, 2):
                        for ix in range(1 + (col & 1), nx, 2):
                            one(ix, iy, iz)
            continue
        P, Q = {1: (1, 2), 2: (0, 2), 3: (0, 1)}[direction]
        for col in (bwd if iback else fwd):
            for iQ in range(1 + (col >> 1), nC[Q], 2):
                for iP in range(1 + (col & 1), nC[P], 2):
                    idx = [None, None, None]
                    idx[P], idx[Q] = iP, iQ
                    one(*idx)



def colour_fixture(emg3d):
    """The 4-/8-colour schedule of the device path replayed with REFERENCE arithmetic.

    The reference's own ``core.gauss_seidel_x/_y/_z`` with ``nu=1`` on the sub-arrays that span the nodes
    ``iP-1..iP+1, iQ-1..iQ+1`` (2 x 2 cells transversally, the full length along the line) performs exactly the one line
    update (iP, iQ): the boundary edges of the sub-grid are read, never written (core.py:572-602, 745-753).  Likewise
    ``core.gauss_seidel`` on 2 x 2 x 2 cells updates the one node (core.py:285-309, 469-474).  Any schedule can therefore
    be replayed.  Schedule: line colour = ((iP-1)&1) + 2*((iQ-1)&1) with (P, Q) = (y, z), (x, z), (x, y) for x-, y-,
    z-lines; odd sweeps (1st, 3rd: the reference's "backward" ones) visit the colours 0,3,2,1, even sweeps 1,3,0,2;
    points: colour = ((ix-1)&1) + 2*((iy-1)&1) + 4*((iz-1)&1), 0..7 in every sweep.  Stored: the result after all
    4*nu (8*nu) colour passes -- the device skips the pass that repeats the previous one's colour, which re-solves
    unchanged systems.  Inputs: the kernel fixtures' (kernels_c128.npz / kernels_f64.npz, written by this script) and
    one ragged odd grid generated here."""
    from emg3d import core, fields, meshes, models

    def replay(nC, e, s, eta, zeta, h, direction, nu, lex=False):
        colour_replay(core, nC, e, s, eta, zeta, h, direction, nu, lex)

    out = {}
    cases = {}
    g_all = {}
    for tag, fname in (('c128', 'kernels_c128.npz'), ('f64', 'kernels_f64.npz')):
        g = np.load(os.path.join(HERE, fname))
        cases[tag] = {k: g[k] for k in ('hx', 'hy', 'hz', 'e', 's', 'eta_x', 'eta_y', 'eta_z', 'zeta')}
        g_all[tag] = g
    # a ragged odd grid: odd cell counts (colours of unequal size, a last colour row / column without partner), generated
    # through the reference's own Model / VolumeModel
    rng = np.random.default_rng(55)
    hx, hy, hz = rng.uniform(20, 60, 7) * 1.2 ** np.arange(7), rng.uniform(20, 60, 5), rng.uniform(20, 60, 9)
    grid = meshes.TensorMesh([hx, hy, hz], origin=np.array([0., 0., 0.]))
    rho = 10 ** rng.uniform(-0.5, 1.5, (3, grid.nC))
    sf = fields.SourceField(grid, freq=0.7)
    vm = models.VolumeModel(grid, models.Model(grid, rho[0], rho[1], rho[2], mu_r=rng.uniform(0.8, 1.5, grid.nC)), sf)
    e = fields.Field(grid, rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE), freq=0.7)
    e.ensure_pec
    s = fields.Field(grid, (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-3, freq=0.7)
    s.ensure_pec
    cases['odd'] = dict(hx=hx, hy=hy, hz=hz, e=np.array(e), s=np.array(s), eta_x=vm.eta_x, eta_y=vm.eta_y,
                        eta_z=vm.eta_z, zeta=vm.zeta)
    for k, v in cases['odd'].items():
        out[f'odd_{k}'] = v
    # a grid with LONG lines along x (70 blocks: several waves per line and two blocks per quad in the scan kernel; the 65...128-block
    # configuration of the affine chain kernel) and lines of 6 / 5 blocks across
    rng = np.random.default_rng(56)
    hx, hy, hz = rng.uniform(20, 60, 70) * 1.02 ** np.abs(np.arange(70) - 35), rng.uniform(20, 60, 6), rng.uniform(20, 60, 5)
    grid = meshes.TensorMesh([hx, hy, hz], origin=np.array([0., 0., 0.]))
    rho = 10 ** rng.uniform(-0.5, 1.5, (3, grid.nC))
    sf = fields.SourceField(grid, freq=2.0)
    vm = models.VolumeModel(grid, models.Model(grid, rho[0], rho[1], rho[2]), sf)
    e = fields.Field(grid, rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE), freq=2.0)
    e.ensure_pec
    s = fields.Field(grid, (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-3, freq=2.0)
    s.ensure_pec
    cases['long'] = dict(hx=hx, hy=hy, hz=hz, e=np.array(e), s=np.array(s), eta_x=vm.eta_x, eta_y=vm.eta_y,
                         eta_z=vm.eta_z, zeta=vm.zeta)
    for k, v in cases['long'].items():
        out[f'long_{k}'] = v
    for tag, c in cases.items():
        nC = (c['hx'].size, c['hy'].size, c['hz'].size)
        args = ((c['eta_x'], c['eta_y'], c['eta_z']), c['zeta'], (c['hx'], c['hy'], c['hz']))
        for direction, name in enumerate(('gs', 'gs_x', 'gs_y', 'gs_z')):
            if tag in g_all:
                ee = c['e'].copy()
                replay(nC, ee, c['s'], *args, direction, 1, lex=True)
                assert np.array_equal(ee, g_all[tag][f'{name}_nu1']), (tag, name)
            for nu in (1, 2, 3):
                ee = c['e'].copy()
                replay(nC, ee, c['s'], *args, direction, nu)
                out[f'{tag}_{name}_colour_nu{nu}'] = ee
                print('colour', tag, name, nu, np.abs(ee).max())
    return out


def colour_solve_fixture(emg3d):
    """Complete multigrid SOLVES in the device's colour ordering with REFERENCE arithmetic: the reference's own `solver.solve` with
    `solver.smoothing` replaced by the sub-grid replay of the colour schedule (`colour_replay`; the dispatch over `lr_dir` incl. the
    two-cell rule is the reference's own, solver.py:784-799) -- every line / node update is the reference's kernel, only their ORDER is
    the device's.  Problem: the 16^3 stretched random tri-axial grid of solves_16.npz.  Stored: field, per-cycle norms, counts."""
    from emg3d import core, solver, fields, meshes, models
    g = np.load(os.path.join(HERE, 'solves_16.npz'))
    grid = meshes.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = models.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = fields.get_source_field(grid, list(g['src']), float(g['freq']))
    assert np.array_equal(np.array(sfield), g['sfield'])

    def smoothing_colour(grid, model, sfield, efield, nu, lr_dir):
        lr = solver._current_lr_dir(lr_dir, grid)
        a = (tuple(int(n) for n in grid.shape_cells), efield.field, sfield.field, (model.eta_x, model.eta_y, model.eta_z),
             model.zeta, grid.h)
        if lr == 0:
            colour_replay(core, *a, 0, nu)
        if lr in [1, 5, 6, 7]:
            colour_replay(core, *a, 1, nu)
        if lr in [2, 4, 6, 7]:
            colour_replay(core, *a, 2, nu)
        if lr in [3, 4, 5, 7]:
            colour_replay(core, *a, 3, nu)
    orig = solver.smoothing
    solver.smoothing = smoothing_colour
    out = {}
    try:
        for name, kw in (('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)),
                         ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True)),
                         ('F_plain', dict(cycle='F', maxit=5)),
                         ('bic_sclr', dict(sslsolver=True, semicoarsening=True, linerelaxation=True))):
            ef, info = solver.solve(grid, model, sfield, return_info=True, verb=1, **kw)
            out[f'{name}_efield'] = np.array(ef)
            out[f'{name}_error_at_cycle'] = info['error_at_cycle']
            out[f'{name}_it'] = np.array([info['it_mg'], info['it_ssl']])
            out[f'{name}_exit'] = np.array(info['exit'])
            print('colour solve', name, info['it_mg'], info['rel_error'], info['exit_message'], flush=True)
        # Laplace domain (real arithmetic: the float64 kernels), s = 2.0
        sf_lap = fields.get_source_field(grid, list(g['src']), -2.0)
        out['lap_sfield'] = np.array(sf_lap)
        ef, info = solver.solve(grid, model, sf_lap, return_info=True, verb=1, cycle='F', semicoarsening=True, linerelaxation=True)
        out['lap_F_sclr_efield'] = np.array(ef)
        out['lap_F_sclr_error_at_cycle'] = info['error_at_cycle']
        out['lap_F_sclr_it'] = np.array([info['it_mg'], info['it_ssl']])
        out['lap_F_sclr_exit'] = np.array(info['exit'])
        print('colour solve lap', info['it_mg'], info['rel_error'], info['exit_message'], flush=True)
    finally:
        solver.smoothing = orig
    return out


def main():
    emg3d = _import_reference()
    big = '--big' in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith('--')]

    def want(name):
        return not only or name in only

    if want('kernels'):
        np.savez_compressed(os.path.join(HERE, 'kernels_c128.npz'),
                            **kernel_fixture(emg3d, np.complex128, 11))
        np.savez_compressed(os.path.join(HERE, 'kernels_f64.npz'),
                            **kernel_fixture(emg3d, np.float64, 12))
    if want('colour'):
        np.savez_compressed(os.path.join(HERE, 'kernels_colour.npz'), **colour_fixture(emg3d))
    if want('colour_solves'):
        np.savez_compressed(os.path.join(HERE, 'solves_16_colour.npz'), **colour_solve_fixture(emg3d))
    if want('source'):
        np.savez_compressed(os.path.join(HERE, 'source_fields.npz'), **source_fixture(emg3d))
    if want('entry'):
        # solver.multigrid fixes level 0's cycmax on entry from the FIRST sc_dir (solver.py:480-485): with
        # semicoarsening=True on 8 x 3 x 3 the first direction (1) has clevel 0, so all later F-cycles visit the coarse
        # levels once.  Inputs + per-cycle norms + field of the reference.
        from emg3d import solver, fields, meshes, models
        rng = np.random.default_rng(3)
        out = {}
        for tag, shape, freq, lr in (('a', (8, 3, 3), -0.5, 1), ('b', (48, 5, 5), 0.1, 4), ('c', (12, 3, 6), 1.0, 7)):
            h = [rng.uniform(20, 60) * 1.1 ** np.abs(np.arange(n) - n / 2 + 0.5) for n in shape]
            origin = np.array([-hh.sum() / 2 for hh in h])
            grid = meshes.TensorMesh(h, origin=origin)
            rho = 10 ** rng.uniform(-0.5, 2.0, grid.nC)
            model = models.Model(grid, rho, property_z=rho * 2)
            src = [1., 2., 0.5, 30., 10.]
            sfield = fields.get_source_field(grid, src, freq)
            opts = dict(cycle='F', semicoarsening=True, linerelaxation=lr, nu_init=0, nu_pre=2, nu_coarse=2, nu_post=2,
                        maxit=3, tol=1e-14)
            ef, info = solver.solve(grid, model, sfield, return_info=True, verb=0, **opts)
            out.update({f'{tag}_hx': h[0], f'{tag}_hy': h[1], f'{tag}_hz': h[2], f'{tag}_origin': origin, f'{tag}_rho': rho,
                        f'{tag}_freq': freq, f'{tag}_lr': lr, f'{tag}_src': np.array(src), f'{tag}_efield': np.array(ef),
                        f'{tag}_error_at_cycle': np.array(info['error_at_cycle'])})
        np.savez_compressed(os.path.join(HERE, 'solves_entry.npz'), **out)
    if want('div'):
        # the diverging case of the reference's tests/test_solver.py:test_solver_heterogeneous: 2**9 x 2 x 2 cells, no
        # pre-smoothing, the point source of the reference's tests/alternatives.py (two non-zero edges)
        sys.path.insert(0, os.path.join(REF, 'tests'))
        import alternatives
        from emg3d import solver, meshes, models
        mesh = meshes.TensorMesh([np.ones(2**9) / np.ones(2**9).sum(), np.ones(2), np.ones(2)], origin=np.array([-0.5, -1, -1]))
        sfield = alternatives.get_source_field(mesh, [0, 0, 0, 0, 0], 1)
        e, info = solver.solve(mesh, models.Model(mesh), sfield, verb=0, nu_pre=0, return_info=True)
        np.savez_compressed(os.path.join(HERE, 'solves_div.npz'), hx=mesh.h[0], hy=mesh.h[1], hz=mesh.h[2],
                            origin=np.array([-0.5, -1, -1.]), sfield=np.array(sfield), freq=1.0,
                            error_at_cycle=np.array(info['error_at_cycle']), it_mg=info['it_mg'],
                            exit_message=str(info['exit_message']), efield=np.array(e))
    if want('gradient'):
        np.savez_compressed(os.path.join(HERE, 'gradient.npz'), **gradient_fixture(emg3d))
    if want('receivers'):
        np.savez_compressed(os.path.join(HERE, 'receivers.npz'), **receivers_fixture(emg3d))
    if want('logs'):
        np.savez_compressed(os.path.join(HERE, 'logs.npz'), **logs_fixture(emg3d))
    if want('receiver_modes'):
        np.savez_compressed(os.path.join(HERE, 'receivers_modes.npz'), **receiver_modes_fixture(emg3d))
    if want('regression'):
        np.savez_compressed(os.path.join(HERE, 'regression.npz'), **regression_fixture(emg3d))
    if want('solves16'):
        kinds = {
            'F_sclr': dict(cycle='F', semicoarsening=True, linerelaxation=True),
            'V_sclr': dict(cycle='V', semicoarsening=True, linerelaxation=True),
            'W_sclr': dict(cycle='W', semicoarsening=True, linerelaxation=True),
            'F_plain': dict(cycle='F', maxit=5),
            'bic_sclr': dict(sslsolver=True, semicoarsening=True, linerelaxation=True),
        }
        np.savez_compressed(os.path.join(HERE, 'solves_16.npz'),
                            **solves_fixture(emg3d, 16, 8, 4, kinds))
    if want('solves_eps'):
        np.savez_compressed(os.path.join(HERE, 'solves_eps.npz'), **solves_eps_fixture(emg3d))
    if big and want('solves32'):
        from emg3d import solver, fields, meshes, models
        h = np.ones(32) * 50.
        grid = meshes.TensorMesh([h, h, h], origin=(-800, -800, -800))
        model = models.Model(grid, 1.)
        sfield = fields.get_source_field(grid, [0, 0, 0, 30, 10], 1.0)
        ef, info = solver.solve(grid, model, sfield, cycle='F', return_info=True, verb=1)
        np.savez_compressed(os.path.join(HERE, 'solves_32.npz'), h=h, sfield=np.array(sfield),
                            smu0=np.array(sfield.smu0), efield=np.array(ef),
                            error_at_cycle=info['error_at_cycle'])


if __name__ == '__main__':
    main()
