"""GPU parity, kernel level: the HIP path (through the C ABI, via the drop-in
``emg3d_amd.core`` / ``emg3d_amd.solver`` sub-routines) against
(a) golden vectors captured from the reference and (b) the CPU oracle on the
same seeded inputs.  Tolerances are relative max-norm on complex128/float64;
the GPU evaluates the same factorisation with re-associated sums."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

TOL = 2e-10   # two-sided elimination + explicit block inverses: rounding differs from the band LDL^T


@pytest.fixture(scope="module")
def em():
    import emg3d_amd
    from emg3d_amd import _lib
    assert _lib.device_count() >= 1
    return emg3d_amd


@pytest.fixture(scope="module", params=["kernels_c128.npz", "kernels_f64.npz"])
def gold(request):
    return load_golden(request.param)


def _grid(em, g):
    return em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])


class _VM:
    def __init__(self, g, p=''):
        self.eta_x, self.eta_y, self.eta_z, self.zeta = (g[p + 'eta_x'], g[p + 'eta_y'], g[p + 'eta_z'], g[p + 'zeta'])
        self.case = 3


def _field(em, grid, arr, freq):
    return em.Field(grid, arr.copy(), freq=freq)


def test_amat_x(em, gold):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    r = _field(em, grid, gold['s'], f)
    e = _field(em, grid, gold['e'], f)
    em.core.amat_x(r.fx, r.fy, r.fz, e.fx, e.fy, e.fz, gold['eta_x'], gold['eta_y'], gold['eta_z'],
                   gold['zeta'], *grid.h)
    assert relerr(r, gold['amat_x_r']) < TOL
    n = em.solver.residual(grid, _VM(gold), _field(em, grid, gold['s'], f), e, True)
    assert abs(n / gold['residual_norm'] - 1) < 1e-12


@pytest.mark.parametrize("name", ['gs', 'gs_x', 'gs_y', 'gs_z'])
@pytest.mark.parametrize("nu", [1, 2, 3])
def test_gauss_seidel_lex_vs_reference(em, gold, name, nu):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    e = _field(em, grid, gold['e'], f)
    s = _field(em, grid, gold['s'], f)
    fn = {'gs': em.core.gauss_seidel, 'gs_x': em.core.gauss_seidel_x, 'gs_y': em.core.gauss_seidel_y,
          'gs_z': em.core.gauss_seidel_z}[name]
    fn(e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, gold['eta_x'], gold['eta_y'], gold['eta_z'], gold['zeta'],
       *grid.h, nu)
    assert relerr(e, gold[f'{name}_nu{nu}']) < TOL


@pytest.mark.parametrize("direction", [0, 1, 2, 3])
def test_gauss_seidel_colour_vs_oracle(em, oracle, gold, direction):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    e = _field(em, grid, gold['e'], f)
    s = _field(em, grid, gold['s'], f)
    em.core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, gold['eta_x'], gold['eta_y'], gold['eta_z'],
                gold['zeta'], *grid.h, 2, order=1)
    eo = gold['e'].copy()
    oracle.gauss_seidel(grid.vnC, eo, gold['s'], gold['eta_x'], gold['eta_y'], gold['eta_z'], gold['zeta'],
                        *grid.h, 2, direction=direction, order=1)
    assert relerr(e, eo) < TOL


@pytest.mark.parametrize("tag,fname", [('c128', 'kernels_c128.npz'), ('f64', 'kernels_f64.npz'), ('odd', None), ('long', None)])
@pytest.mark.parametrize("direction,name", [(0, 'gs'), (1, 'gs_x'), (2, 'gs_y'), (3, 'gs_z')])
@pytest.mark.parametrize("nu", [1, 2, 3])
def test_gauss_seidel_colour_vs_reference(em, tag, fname, direction, name, nu):
    """The colour ordering (what `bench.py` times), INCLUDING the skipped turn-around pass (csrc/mg.hpp: skip_idempotent),
    against the schedule replayed with the reference's own kernels on sub-grids (kernels_colour.npz)."""
    col = load_golden('kernels_colour.npz')
    if fname is None:
        g = {k: col[f'{tag}_{k}'] for k in ('hx', 'hy', 'hz', 'e', 's', 'eta_x', 'eta_y', 'eta_z', 'zeta')}
        g['origin'] = np.zeros(3)
        freq = {'odd': 0.7, 'long': 2.0}[tag]
    else:
        g = load_golden(fname)
        freq = float(g['freq'])
    grid = _grid(em, g)
    e = _field(em, grid, g['e'], freq)
    s = _field(em, grid, g['s'], freq)
    em.core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, g['eta_x'], g['eta_y'], g['eta_z'], g['zeta'],
                *grid.h, nu, order=1)
    assert relerr(e, col[f'{tag}_{name}_colour_nu{nu}']) < TOL


@pytest.mark.parametrize("lr_dir", range(8))
def test_smoothing_dispatch(em, gold, lr_dir):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    e = _field(em, grid, gold['e'], f)
    s = _field(em, grid, gold['s'], f)
    em.solver.smoothing(grid, _VM(gold), s, e, 2, lr_dir)
    assert relerr(e, gold[f'smoothing_lr{lr_dir}']) < TOL


@pytest.mark.parametrize("sc_dir", range(7))
def test_restriction(em, gold, sc_dir):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    s = _field(em, grid, gold['s'], f)
    res = _field(em, grid, gold['res'], f)
    cgrid, cmodel, cs, ce = em.solver.restriction(grid, _VM(gold), s, res, sc_dir)
    p = f'restrict{sc_dir}_'
    for a, c in enumerate('xyz'):
        np.testing.assert_allclose(cgrid.h[a], gold[p + 'ch' + c], rtol=1e-14)
    for k in ('eta_x', 'eta_y', 'eta_z', 'zeta'):
        assert relerr(getattr(cmodel, k), gold[p + k]) < 1e-14
    assert relerr(cs, gold[p + 'csfield']) < TOL
    assert not np.asarray(ce).any()


@pytest.mark.parametrize("sc_dir", range(7))
def test_prolongation(em, gold, sc_dir):
    grid = _grid(em, gold)
    f = float(gold['freq'])
    rx = 1 if sc_dir in [1, 5, 6] else 2
    ry = 1 if sc_dir in [2, 4, 6] else 2
    rz = 1 if sc_dir in [3, 4, 5] else 2
    cgrid = em.TensorMesh([np.diff(grid.nodes_x[::rx]), np.diff(grid.nodes_y[::ry]),
                           np.diff(grid.nodes_z[::rz])], grid.origin)
    e = _field(em, grid, gold['e'], f)
    ce = _field(em, cgrid, gold[f'prolong{sc_dir}_ce'], f)
    em.solver.prolongation(grid, e, cgrid, ce, sc_dir)
    assert relerr(e, gold[f'prolong{sc_dir}_e']) < TOL


def test_solve_kat(em):
    """core.solve: 6x6 and banded known answers vs numpy (reference
    tests/test_core.py:163-222)."""
    rng = np.random.default_rng(3)
    for dtype in (np.float64, np.complex128):
        for n in (6, 21):
            full = np.zeros((n, n), dtype=dtype)
            amat = np.zeros(6 * n, dtype=dtype)
            for j in range(n):
                for i in range(j, min(n, j + 6)):
                    v = rng.standard_normal() + (1j * rng.standard_normal() if dtype == np.complex128 else 0)
                    if i == j:
                        v += 8
                    amat[i + 5 * j] = v
                    full[i, j] = v
                    full[j, i] = v
            b = rng.standard_normal(n).astype(dtype)
            x = b.copy()
            em.core.solve(amat.copy(), x)
            np.testing.assert_allclose(x, np.linalg.solve(full, b), rtol=1e-11)


def test_blocks_to_amat_vs_oracle(em, oracle):
    n = 4
    rng = np.random.default_rng(9)
    for dtype in (np.float64, np.complex128):
        a1 = np.zeros(6 * (5 * n - 4), dtype=dtype); b1 = np.zeros(5 * n - 4, dtype=dtype)
        a2 = a1.copy(); b2 = b1.copy()
        for im in range(n):
            middle = rng.standard_normal(25).astype(dtype)
            left = rng.standard_normal(25)
            rhs = rng.standard_normal(5).astype(dtype)
            em.core.blocks_to_amat(a1, b1, middle, left, rhs, im, n)
            oracle.blocks_to_amat(a2, b2, middle, left, rhs, im, n)
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2)


@pytest.mark.parametrize("vnC", [(32, 16, 8), (8, 24, 12), (6, 10, 40)])
@pytest.mark.parametrize("order", [0, 1])
def test_sweeps_vs_oracle_ragged(em, oracle, vnC, order):
    """Non-cubic / non-power-of-two grids, every smoother, both orderings."""
    rng = np.random.default_rng(sum(vnC) + order)
    h = [rng.uniform(10, 80, n) for n in vnC]
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    eta = [np.asfortranarray(-1j * 8e-6 * vol * 10 ** rng.uniform(-1.5, 0.5, grid.vnC)) for _ in range(3)]
    zeta = np.asfortranarray(vol / rng.uniform(0.9, 1.3, grid.vnC))
    e0 = em.Field(grid, (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)), freq=1.)
    e0.ensure_pec
    s = em.Field(grid, 1e-6 * (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)), freq=1.)
    s.ensure_pec
    for direction in (0, 1, 2, 3):
        e = e0.copy()
        em.core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, *eta, zeta, *grid.h, 2, order=order)
        eo = np.array(e0)
        oracle.gauss_seidel(grid.vnC, eo, np.array(s), *eta, zeta, *grid.h, 2, direction=direction, order=order)
        assert relerr(e, eo) < TOL, (direction, relerr(e, eo))


def test_cabi_argument_validation():
    """Error behaviour of the C ABI (include/emg3d_hip.h): int status, < 0 for invalid arguments, no
    crash on NULL handles; the Python shim turns every non-zero status into HipLibraryError."""
    import ctypes
    from emg3d_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(None)
    d = ctypes.c_double()
    # NULL handle
    assert lib.emg3d_mg_cycle(null, 0, 0, ctypes.byref(d)) == -1
    assert lib.emg3d_mg_set_sfield(null, None) == -1
    assert lib.emg3d_mg_vec_alloc(null, 3) == -1
    assert lib.emg3d_mg_nE(null) == -1
    assert not lib.emg3d_mg_efield_devptr(null)
    lib.emg3d_mg_destroy(null)                              # no-op
    # grids smaller than 2 cells per axis
    h = np.ones(4)
    eta = np.ones(16, dtype=np.complex128)
    zeta = np.ones(16)
    out = ctypes.c_void_p()
    assert lib.emg3d_mg_create(ctypes.byref(out), 1, 1, 4, 4, _lib.ptr(h), _lib.ptr(h), _lib.ptr(h), None,
                               _lib.ptr(eta), None, None, _lib.ptr(zeta), 0) == -2
    assert lib.emg3d_mg_create(None, 1, 4, 4, 4, _lib.ptr(h), _lib.ptr(h), _lib.ptr(h), None,
                               _lib.ptr(eta), None, None, _lib.ptr(zeta), 0) == -1
    # a valid 4x4x4 handle and invalid arguments on it
    eta = np.full(64, -1e-3j, dtype=np.complex128)
    zeta = np.ones(64)
    assert lib.emg3d_mg_create(ctypes.byref(out), 1, 4, 4, 4, _lib.ptr(h), _lib.ptr(h), _lib.ptr(h), None,
                               _lib.ptr(eta), None, None, _lib.ptr(zeta), 0) == 0
    try:
        cl = (ctypes.c_int * 4)(1, 1, 1, 1)
        assert lib.emg3d_mg_set_params(out, ord('X'), 0, 2, 1, 2, cl, 1) == -2      # cycle
        assert lib.emg3d_mg_set_params(out, ord('F'), 0, 2, 1, 2, cl, 5) == -2      # ordering
        assert lib.emg3d_mg_set_params(out, ord('F'), 0, 2, 1, 2, cl, 1) == 0
        assert lib.emg3d_mg_cycle(out, 4, 0, ctypes.byref(d)) == -2                 # sc_dir
        assert lib.emg3d_mg_cycle(out, 0, 8, ctypes.byref(d)) == -2                 # lr_dir
        assert lib.emg3d_mg_prepare(out, -1, 0) == -2
        assert lib.emg3d_mg_smooth(out, -1, 0) == -2
        assert lib.emg3d_mg_cycles(out, 0, None, 1, None, 1, None) == -2
        assert lib.emg3d_mg_vec_alloc(out, 1000) == -2
        assert lib.emg3d_mg_vec_copy(out, 0, 1) == -2                                # not allocated
        assert lib.emg3d_mg_vec_dot(out, 0, 0, None) == -2
        assert lib.emg3d_mg_set_sfield_vector(out, None, 1.0, 0.0) == -2
        f = ctypes.c_float()
        assert lib.emg3d_mg_time_sweep(out, 9, 1, ctypes.byref(f)) == -2
        assert lib.emg3d_mg_cycle(out, 0, 0, ctypes.byref(d)) == 0 and d.value == 0.0    # zero source: zero residual
        with pytest.raises(_lib.HipLibraryError, match="invalid argument"):
            _lib.check(lib.emg3d_mg_cycle(out, 9, 0, ctypes.byref(d)), "emg3d_mg_cycle")
    finally:
        lib.emg3d_mg_destroy(out)
    # tier 1
    assert lib.emg3d_gauss_seidel(1, 7, 4, 4, 4, None, None, None, None, None, None, None, None, None, 1, 0) == -2
    assert lib.emg3d_gauss_seidel(1, 1, 4, 4, 4, None, None, None, None, None, None, None, None, None, 1, 3) == -2


def test_line_smoothers_equal_point_smoother_like_the_reference(em):
    """reference tests/test_core.py:test_gauss_seidel: on grids that are two cells wide along the line direction the line
    smoothers gauss_seidel_x/_y/_z must reproduce the point smoother gauss_seidel (same inputs: two solver iterations as
    starting field, nu = 2) -- here through the drop-in emg3d_amd.core on the device."""
    src = [0., 0., 0., 45., 45.]
    freq = 0.9
    nu = 2
    for lr_dir in range(1, 4):
        nx, ny, nz = [1, 4, 4][lr_dir - 1], [4, 1, 4][lr_dir - 1], [4, 4, 1][lr_dir - 1]
        hx = em.meshes.stretched_widths(0, nx, 80., 1.1)
        hy = em.meshes.stretched_widths(0, ny, 100., 1.3)
        hz = em.meshes.stretched_widths(0, nz, 200., 1.2)
        grid = em.TensorMesh([hx, hy, hz], np.array([-hx.sum() / 2, -hy.sum() / 2, -hz.sum() / 2]))
        model = em.Model(grid, np.arange(grid.nC) + 1., 0.5 * np.arange(grid.nC) + 1., 2. * np.arange(grid.nC) + 1.)
        sfield = em.get_source_field(grid, src, freq)
        vmodel = em.VolumeModel(grid, model, sfield)
        efield = em.solve(grid, model, sfield, maxit=2, verb=0, ordering='lex')
        inp = (sfield.fx, sfield.fy, sfield.fz, vmodel.eta_x, vmodel.eta_y, vmodel.eta_z, vmodel.zeta, grid.h[0], grid.h[1],
               grid.h[2], nu)
        cfield = em.Field(grid, np.array(efield).copy(), freq=freq)
        em.core.gauss_seidel(cfield.fx, cfield.fy, cfield.fz, *inp)
        line = (em.core.gauss_seidel_x, em.core.gauss_seidel_y, em.core.gauss_seidel_z)[lr_dir - 1]
        line(efield.fx, efield.fy, efield.fz, *inp)
        np.testing.assert_allclose(np.array(efield), np.array(cfield), rtol=1e-7, atol=1e-20)


def test_smoothing_restriction_residual_like_the_reference(em):
    """The reference's tests/test_solver.py: test_smoothing (solver.smoothing == the core smoothers for every lr_dir, with
    the 2-cell dimension moved through x, y, z), test_restriction (known values, prolongation of a constant) and
    test_residual (solver.residual == sfield - amat_x, norm included), through emg3d_amd.solver / emg3d_amd.core."""
    from emg3d_amd import core, solver
    get_h = em.meshes.stretched_widths
    nu = 2
    widths = [np.ones(2) * 100, get_h(10, 27, 10., 1.1), get_h(2, 1, 50., 1.2)]
    origin = [-w.sum() / 2 for w in widths]
    src = [0., -10., -10., 43., 13.]
    for xyz in range(3):
        grid = em.TensorMesh([widths[xyz % 3], widths[(xyz + 1) % 3], widths[(xyz + 2) % 3]],
                             origin=np.array([origin[xyz % 3], origin[(xyz + 1) % 3], origin[(xyz + 2) % 3]]))
        x = np.arange(1, grid.vnC[0] + 1) * 2
        y = 1 / np.arange(1, grid.vnC[1] + 1)
        z = np.arange(1, grid.vnC[2] + 1)[::-1] / 10
        property_x = np.outer(np.outer(x, y), z).ravel()
        freq = 0.319
        model = em.Model(grid, property_x, 0.8 * property_x, 2 * property_x)
        sfield = em.get_source_field(grid, src, freq)
        vmodel = em.VolumeModel(grid, model, sfield)
        field = em.solve(grid, model, sfield, maxit=2, verb=0, ordering='lex')
        inp = (sfield.fx, sfield.fy, sfield.fz, vmodel.eta_x, vmodel.eta_y, vmodel.eta_z, vmodel.zeta, grid.h[0], grid.h[1],
               grid.h[2], nu)
        seq = {0: [core.gauss_seidel], 1: [core.gauss_seidel_x], 2: [core.gauss_seidel_y], 3: [core.gauss_seidel_z],
               4: [core.gauss_seidel_y, core.gauss_seidel_z], 5: [core.gauss_seidel_x, core.gauss_seidel_z],
               6: [core.gauss_seidel_x, core.gauss_seidel_y], 7: [core.gauss_seidel_x, core.gauss_seidel_y, core.gauss_seidel_z]}
        for lr_dir in range(8):
            # (the reference's test wraps the SAME buffer twice -- Field(grid, field) does not copy -- and so compares a
            # field with itself; here the fields are copies, and the expected sequence is the one of the direction that
            # solver.smoothing really uses: it drops line relaxation along 2-cell dimensions, solver.py:790)
            efield = em.Field(grid, np.array(field).copy(), freq=freq)
            for fn in seq[int(solver._current_lr_dir(lr_dir, grid))]:
                fn(efield.fx, efield.fy, efield.fz, *inp)
            ofield = em.Field(grid, np.array(field).copy(), freq=freq)
            solver.smoothing(grid, vmodel, sfield, ofield, nu, lr_dir)
            np.testing.assert_allclose(np.array(efield), np.array(ofield), rtol=1e-7, atol=1e-20)

    # test_restriction
    grid = em.TensorMesh([np.ones(4) * 100, np.ones(4) * 100, np.ones(4) * 100], origin=np.zeros(3))
    model = em.Model(grid, 1., 1., 1., 1.)
    sfield = em.get_source_field(grid, [150., 150., 150., 0., 45.], 1.)
    vmodel = em.VolumeModel(grid, model, sfield)
    rx = np.arange(sfield.fx.size, dtype=np.complex128).reshape(sfield.fx.shape)
    ry = np.arange(sfield.fy.size, dtype=np.complex128).reshape(sfield.fy.shape)
    rz = np.arange(sfield.fz.size, dtype=np.complex128).reshape(sfield.fz.shape)
    rr = em.Field(rx, ry, rz)
    cgrid, cmodel, csfield, cefield = solver.restriction(grid, vmodel, sfield, rr, sc_dir=0)
    np.testing.assert_allclose(csfield.fx[:, 1:-1, 1], np.array([[196. + 0.j], [596. + 0.j]]))
    np.testing.assert_allclose(csfield.fy[1:-1, :, 1], np.array([[356. + 0.j, 436. + 0.j]]))
    np.testing.assert_allclose(csfield.fz[1:-1, 1:-1, :], np.array([[[388. + 0.j, 404. + 0.j]]]))
    assert cgrid.vnN[0] == cgrid.vnN[1] == cgrid.vnN[2] == 3
    assert cmodel.eta_x[0, 0, 0] / 8. == vmodel.eta_x[0, 0, 0]
    for a in range(3):
        assert np.sum(grid.h[a]) == np.sum(cgrid.h[a])
    efield = em.Field(grid)
    cefield += np.pi
    solver.prolongation(grid, efield, cgrid, cefield, sc_dir=0)
    assert np.all(efield.fx[:, 1:-1, 1:-1] == np.pi) and np.all(efield.fy[1:-1, :, 1:-1] == np.pi)
    assert np.all(efield.fz[1:-1, 1:-1, :] == np.pi)

    # test_residual
    grid = em.TensorMesh([get_h(4, 2, 20., 1.2), np.ones(16) * 200, np.ones(2) * 25], origin=np.zeros(3))
    x = np.arange(1, grid.vnC[0] + 1) * 2
    y = 1 / np.arange(1, grid.vnC[1] + 1)
    z = np.arange(1, grid.vnC[2] + 1)[::-1] / 10
    property_x = np.outer(np.outer(x, y), z).ravel()
    model = em.Model(grid, property_x, 0.8 * property_x, 2 * property_x)
    sfield = em.get_source_field(grid, [90., 1600., 25., 45., 45.], 0.319)
    vmodel = em.VolumeModel(grid, model, sfield)
    efield = em.solve(grid, model, sfield, maxit=2, verb=0, ordering='lex')
    rfield = sfield.copy()
    core.amat_x(rfield.fx, rfield.fy, rfield.fz, efield.fx, efield.fy, efield.fz, vmodel.eta_x, vmodel.eta_y, vmodel.eta_z,
                vmodel.zeta, grid.h[0], grid.h[1], grid.h[2])
    out = solver.residual(grid, vmodel, sfield, efield)
    outnorm = solver.residual(grid, vmodel, sfield, efield, True)
    np.testing.assert_allclose(np.array(out), np.array(rfield), rtol=1e-12, atol=1e-25)
    np.testing.assert_allclose(outnorm, np.linalg.norm(np.array(out)), rtol=1e-12)


def test_restrict_like_the_reference(em):
    """reference tests/test_core.py: test_restrict (all seven sc_dir cases on a regular 6^3 grid: the restricted field
    conserves the sum of the fine field, the components are multiples of each other) and test_restrict_weights (Equation 9
    of Mulder 2006) through emg3d_amd.core."""
    from emg3d_amd import core
    h = np.array([1., 1, 1, 1, 1, 1])
    fgrid = em.TensorMesh([h, h, h], origin=np.array([-3., -3, -3]))
    ffield = em.Field(fgrid)
    ffield.fx[:, :, :] = 1
    ffield.fy[:, :, :] = 2
    ffield.fz[:, :, :] = 4
    ffield.ensure_pec
    nN = fgrid.vnC[0] + 1
    fw = (np.zeros(nN), np.ones(nN), np.zeros(nN))
    cgrid0 = em.TensorMesh([np.diff(fgrid.nodes_x[::2]), np.diff(fgrid.nodes_y[::2]), np.diff(fgrid.nodes_z[::2])], fgrid.origin)
    w = core.restrict_weights(fgrid.nodes_x, fgrid.cell_centers_x, fgrid.h[0], cgrid0.nodes_x, cgrid0.cell_centers_x, cgrid0.h[0])
    c2 = lambda a: np.diff(a[::2])
    cases = {0: ([c2(fgrid.nodes_x), c2(fgrid.nodes_y), c2(fgrid.nodes_z)], (w, w, w)),
             1: ([fgrid.h[0], c2(fgrid.nodes_y), c2(fgrid.nodes_z)], (fw, w, w)),
             2: ([c2(fgrid.nodes_x), fgrid.h[1], c2(fgrid.nodes_z)], (w, fw, w)),
             3: ([c2(fgrid.nodes_x), c2(fgrid.nodes_y), fgrid.h[2]], (w, w, fw)),
             4: ([c2(fgrid.nodes_x), fgrid.h[1], fgrid.h[2]], (w, fw, fw)),
             5: ([fgrid.h[0], c2(fgrid.nodes_y), fgrid.h[2]], (fw, w, fw)),
             6: ([fgrid.h[0], fgrid.h[1], c2(fgrid.nodes_z)], (fw, fw, w))}
    for sc_dir, (ch, ws) in cases.items():
        cgrid = em.TensorMesh(ch, fgrid.origin)
        cfield = em.Field(cgrid)
        core.restrict(cfield.fx, cfield.fy, cfield.fz, ffield.fx, ffield.fy, ffield.fz, *ws, sc_dir)
        assert cfield.fx.sum() == ffield.fx.sum() and cfield.fy.sum() == ffield.fy.sum() and cfield.fz.sum() == ffield.fz.sum()
        if sc_dir in (0, 3, 6):
            np.testing.assert_allclose(cfield.fx[0, :, :] * 2, cfield.fy[:, 0, :])
        if sc_dir in (0, 1, 4):
            np.testing.assert_allclose(cfield.fy[:, 0, :] * 2, cfield.fz[:, :, 0])
        if sc_dir in (2, 5):
            np.testing.assert_allclose(cfield.fx[0, :, :].T * 4, cfield.fz[:, :, 0])
    # restrict_weights, Equation 9 of [Muld06]
    edges = np.array([0., 500, 1200, 2000, 3000])
    width = edges[1:] - edges[:-1]
    centr = edges[:-1] + width / 2
    c_edges = edges[::2]
    c_width = c_edges[1:] - c_edges[:-1]
    c_centr = c_edges[:-1] + c_width / 2
    wl, w0, wr = core.restrict_weights(edges, centr, width, c_edges, c_centr, c_width)
    np.testing.assert_allclose([350 / 250, 250 / 600, 400 / 900], wl)
    np.testing.assert_allclose([1., 1., 1.], w0)
    np.testing.assert_allclose([350 / 600, 500 / 900, 400 / 500], wr)


def _sweeps_against_oracle(oracle, dtype, shape, kernel_of_dir, elsewhere_not="k_line_sweep_tha"):
    """Two colour-ordered sweeps per direction on random fields / model / widths against the oracle's smoother; kernel_of_dir
    maps a direction to the prefix the selected kernel's name must have (None: anything but `elsewhere_not`)."""
    from types import SimpleNamespace
    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG, MGParameters
    rng = np.random.default_rng(sum(shape))
    cplx = dtype == np.complex128
    h = [rng.uniform(0.5, 2, n) for n in shape]
    grid = em.TensorMesh(h, origin=(0, 0, 0))

    def rnd(n):
        a = rng.standard_normal(n)
        return a + 1j * rng.standard_normal(n) if cplx else a

    if cplx:
        eta = [np.asfortranarray(rng.uniform(0.5, 2, shape) * 0.3j) for _ in range(3)]
        kw = dict(freq=1.)
    else:
        eta = [np.asfortranarray(-rng.uniform(0.5, 2, shape)) for _ in range(3)]
        kw = dict(freq=-1.)
    zeta = np.asfortranarray(rng.uniform(0.5, 2, shape))
    s = em.Field(grid, rnd(grid.nE), **kw)
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC,
                       ordering='colour')
    with DeviceMG(grid, SimpleNamespace(eta_x=eta[0], eta_y=eta[1], eta_z=eta[2], zeta=zeta), np.dtype(dtype)) as dev:
        dev.set_params(var)
        dev.set_sfield(s)
        for direction in (1, 2, 3):
            e0 = em.Field(grid, rnd(grid.nE), **kw)
            dev.set_efield(e0)
            dev.smooth(2, direction)
            e = dev.get_efield()
            name = dev.last_sweep_kernel()
            want = kernel_of_dir.get(direction)
            assert name.startswith(want) if want else not name.startswith(elsewhere_not), (direction, name)
            eo = np.array(e0)
            oracle.gauss_seidel(grid.vnC, eo, np.array(s), *eta, zeta, *grid.h, 2, direction=direction, order=1)
            assert relerr(e, eo) < 1e-10, (direction, name, relerr(e, eo))


@pytest.mark.parametrize("dtype", [np.complex128, np.float64])
@pytest.mark.parametrize("shape,rs_dirs", [((64, 70, 66), (1, 3)), ((40, 80, 80), (1,)), ((34, 67, 69), (1,)), ((72, 47, 66), (2,)),
                                           ((70, 68, 51), (3,)), ((33, 68, 68), (1,)), ((48, 70, 68), (1,)), ((64, 48, 44), ()),
                                           ((128, 70, 66), (1,)), ((101, 68, 72), (1, 2, 3))])
def test_two_sided_affine_kernel_with_helper_waves(oracle, dtype, shape, rs_dirs):
    """k_line_sweep_tha -- the mid levels of a cycle (lines of 33 ... 64 blocks and >= 1100 lines per colour; lines of up to 128
    blocks while a colour has <= 2048 lines): the two-sided line
    solve on the mirrored factorisation with the recurrences in affine form; helper waves form each step's coefficients from
    model, factor, neighbour lines and source and hand them to the two chain waves of a workgroup through a ring in LDS
    (counters, release / acquire at workgroup scope), the chain waves meet at the middle of the line through counters too.
    Two colour-ordered sweeps per direction against the oracle's smoothers; x-, y- and z-lines, ragged last workgroups, odd
    and even line lengths (step counts 0, 1, 2 mod the helpers' stride of 3), both dtypes.  Directions outside the kernel's
    range check the neighbours' parity."""
    _sweeps_against_oracle(oracle, dtype, shape, {d: ("k_line_sweep_tha" if d in rs_dirs else None) for d in (1, 2, 3)})



@pytest.mark.parametrize("freq", [1.0, -3.0])
def test_affine_kernel_on_a_small_level_0(oracle, freq):
    """k_line_sweep_tha where a whole grid is a 'mid level': 40 x 80 x 80 cells, x-lines.  Level 0 of a model without mu_r forms
    zeta from the cell widths (LineArgs::zsep) and skips the source loads of workgroups whose lines carry no source
    (LineArgs::sflag, dipole source): two sweeps against the oracle's smoother, frequency and Laplace domain."""
    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG, MGParameters
    rng = np.random.default_rng(5)
    h = [rng.uniform(40, 60, n) for n in (40, 80, 80)]
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    model = em.Model(grid, rng.uniform(0.5, 5, grid.vnC), 2., 3.)
    sf = em.get_source_field(grid, [h[0].sum() / 2, h[1].sum() / 2, h[2].sum() / 2, 10, 5], freq)
    vm = em.VolumeModel(grid, model, sf)
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC,
                       ordering='colour')
    e0 = rng.standard_normal(grid.nE) * 1e-9 + (1j * rng.standard_normal(grid.nE) * 1e-9 if freq > 0 else 0)
    with DeviceMG(grid, vm, sf.dtype) as dev:
        dev.set_params(var)
        dev.set_sfield(sf)
        dev.set_efield(em.Field(grid, e0.astype(sf.dtype), freq=freq))
        dev.smooth(2, 1)
        e = dev.get_efield()
        assert dev.last_sweep_kernel().startswith("k_line_sweep_tha"), dev.last_sweep_kernel()
    eo = e0.astype(sf.dtype)
    oracle.gauss_seidel(grid.vnC, eo, np.array(sf), np.asfortranarray(vm.eta_x), np.asfortranarray(vm.eta_y),
                        np.asfortranarray(vm.eta_z), np.asfortranarray(vm.zeta), *grid.h, 2, direction=1, order=1)
    assert relerr(e, eo) < 1e-10, relerr(e, eo)
