"""Pin the ORACLE's multigrid cycle (oracle/oracle.py) against the reference:
(i) the golden fields of the reference's own tests/data/regression.npz
    (re-exported in tests/golden/regression.npz), and
(ii) fields + per-cycle error traces captured by running the reference in the
     build container (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import assert_norms_close, load_golden, relerr


def _setup(orc, g, prefix):
    mesh = orc.Mesh([g[f'{prefix}_hx'], g[f'{prefix}_hy'], g[f'{prefix}_hz']], g[f'{prefix}_origin'])
    sfield = g[f'{prefix}_sfield']
    smu0 = g[f'{prefix}_smu0_here']
    vol = mesh.cell_volumes
    px, py, pz = (np.broadcast_to(g[f'{prefix}_property_{c}'].reshape(-1, order='F')
                                  if g[f'{prefix}_property_{c}'].ndim == 3 else g[f'{prefix}_property_{c}'],
                                  (mesh.nC,)).reshape(mesh.vnC, order='F') for c in 'xyz')
    eta = [np.asfortranarray(smu0 * vol / p) for p in (px, py, pz)]
    assert relerr(eta[0], g[f'{prefix}_eta_x_here']) < 1e-14
    return mesh, orc.VModel(eta[0], eta[1], eta[2], np.asfortranarray(vol)), sfield


@pytest.mark.parametrize("key,kw", [('F', {}), ('W', {'cycle': 'W'}), ('V', {'cycle': 'V'}),
                                    ('bic', {'sslsolver': True})])
def test_regression_res(oracle, key, kw):
    g = load_golden("regression.npz")
    mesh, model, sfield = _setup(oracle, g, 'res')
    e, info = oracle.solve(mesh, model, sfield.copy(), **kw)
    # same environment (same mu_0): tight
    assert relerr(e, g[f'res_{key}_here']) < 1e-9
    np.testing.assert_allclose(info['error_at_cycle'], g[f'res_{key}_error_at_cycle'], rtol=1e-6)
    assert info['it_mg'] == g[f'res_{key}_it'][0] and info['it_ssl'] == g[f'res_{key}_it'][1]
    # the reference's 2020 golden (CODATA mu_0 differs): its own bar rtol 1e-7..1e-9
    assert relerr(e, g[f'res_{key}_golden']) < 5e-9


def test_regression_reg2(oracle):
    g = load_golden("regression.npz")
    mesh, model, sfield = _setup(oracle, g, 'reg_2')
    kw = {k: g[f'reg_2_inp_{k}'].item() for k in ('semicoarsening', 'linerelaxation', 'tol', 'maxit',
                                                  'nu_init', 'nu_pre', 'nu_coarse', 'nu_post', 'clevel')}
    e, info = oracle.solve(mesh, model, sfield.copy(), **kw)
    assert relerr(e, g['reg_2_here']) < 1e-9
    np.testing.assert_allclose(info['error_at_cycle'], g['reg_2_error_at_cycle'], rtol=1e-6)
    assert relerr(e, g['reg_2_golden']) < 5e-9


@pytest.mark.parametrize("key,kw", [('F', {}), ('bic', {'sslsolver': True})])
def test_regression_lap(oracle, key, kw):
    """Laplace domain: float64 instantiation of every kernel."""
    g = load_golden("regression.npz")
    mesh, model, sfield = _setup(oracle, g, 'lap')
    assert sfield.dtype == np.float64
    e, info = oracle.solve(mesh, model, sfield.copy(), **kw)
    assert relerr(e, g[f'lap_{key}_here']) < 1e-8
    np.testing.assert_allclose(info['error_at_cycle'], g[f'lap_{key}_error_at_cycle'], rtol=1e-5)
    assert relerr(e, g[f'lap_{key}_golden']) < 1e-8


@pytest.mark.parametrize("name,kw", [
    ('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)),
    ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True)),
    ('W_sclr', dict(cycle='W', semicoarsening=True, linerelaxation=True)),
    ('F_plain', dict(cycle='F', maxit=5)),
    ('bic_sclr', dict(sslsolver=True, semicoarsening=True, linerelaxation=True)),
])
def test_solves_16(oracle, name, kw):
    g = load_golden("solves_16.npz")
    mesh = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    vol = mesh.cell_volumes
    rho = g['rho_b'].reshape(mesh.vnC, order='F')
    eta = [np.asfortranarray(g['smu0'] * vol / (f * rho)) for f in (1, 2, 3)]
    assert relerr(eta[0], g['eta_x']) < 1e-14
    model = oracle.VModel(eta[0], eta[1], eta[2], np.asfortranarray(vol))
    e, info = oracle.solve(mesh, model, g['sfield'].copy(), **kw)
    assert info['it_mg'] == g[f'{name}_it'][0] and info['it_ssl'] == g[f'{name}_it'][1]
    assert info['exit'] == int(g[f'{name}_exit'])
    np.testing.assert_allclose(info['error_at_cycle'], g[f'{name}_error_at_cycle'], rtol=1e-6)
    assert relerr(e, g[f'{name}_efield']) < 1e-9


@pytest.mark.parametrize("name,kw", [
    ('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)),
    ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True)),
    ('F_plain', dict(cycle='F', maxit=5)),
    ('bic_sclr', dict(sslsolver=True, semicoarsening=True, linerelaxation=True)),
])
def test_colour_solves_vs_reference_arithmetic(oracle, name, kw):
    """Whole solves in the colour ordering against the reference's own `solver.solve` whose smoothing calls were replaced by the
    sub-grid replay of the device's colour schedule (tests/golden/solves_16_colour.npz: every line / node update the reference's
    kernel, in the device's order): the oracle's `order=1` twin at cycle level -- counts, exit status, per-cycle norms, field."""
    g = load_golden("solves_16.npz")
    c = load_golden("solves_16_colour.npz")
    mesh = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    vol = mesh.cell_volumes
    rho = g['rho_b'].reshape(mesh.vnC, order='F')
    eta = [np.asfortranarray(g['smu0'] * vol / (f * rho)) for f in (1, 2, 3)]
    model = oracle.VModel(eta[0], eta[1], eta[2], np.asfortranarray(vol))
    e, info = oracle.solve(mesh, model, g['sfield'].copy(), order=1, **kw)
    assert info['it_mg'] == c[f'{name}_it'][0] and info['it_ssl'] == c[f'{name}_it'][1] and info['exit'] == int(c[f'{name}_exit'])
    np.testing.assert_allclose(info['error_at_cycle'], c[f'{name}_error_at_cycle'], rtol=1e-6)
    assert relerr(e, c[f'{name}_efield']) < 1e-9


def test_colour_solve_laplace_vs_reference_arithmetic(oracle):
    """The same in the Laplace domain (s = 2: float64 arithmetic throughout)."""
    g = load_golden("solves_16.npz")
    c = load_golden("solves_16_colour.npz")
    mesh = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    vol = mesh.cell_volumes
    rho = g['rho_b'].reshape(mesh.vnC, order='F')
    import emg3d_amd as em
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    sf = em.SourceField(grid, c['lap_sfield'].copy(), freq=-2.0)
    vm = em.VolumeModel(grid, em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b']), sf)
    model = oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)
    e, info = oracle.solve(mesh, model, c['lap_sfield'].copy(), order=1, cycle='F', semicoarsening=True, linerelaxation=True)
    assert e.dtype == np.float64
    assert info['it_mg'] == c['lap_F_sclr_it'][0] and info['exit'] == int(c['lap_F_sclr_exit'])
    np.testing.assert_allclose(info['error_at_cycle'], c['lap_F_sclr_error_at_cycle'], rtol=1e-6)
    assert relerr(e, c['lap_F_sclr_efield']) < 1e-9


def test_colour_ordering_converges_to_same_field(oracle):
    """4-colour smoothing is a different smoother: more cycles, same
    solution to the tolerance."""
    g = load_golden("solves_16.npz")
    mesh = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    vol = mesh.cell_volumes
    rho = g['rho_b'].reshape(mesh.vnC, order='F')
    eta = [np.asfortranarray(g['smu0'] * vol / (f * rho)) for f in (1, 2, 3)]
    model = oracle.VModel(eta[0], eta[1], eta[2], np.asfortranarray(vol))
    e, info = oracle.solve(mesh, model, g['sfield'].copy(), cycle='F', semicoarsening=True,
                           linerelaxation=True, order=1, tol=1e-8)
    assert info['exit'] == 0
    e_lex, _ = oracle.solve(mesh, model, g['sfield'].copy(), cycle='F', semicoarsening=True,
                            linerelaxation=True, order=0, tol=1e-8)
    assert relerr(e, e_lex) < 1e-6


def test_config_c1_32cubed(oracle):
    """BASELINE config C1 (plumbing): 32^3, h = 50 m, 1 Ohm-m isotropic fullspace, 1 Hz, F-cycle, no
    semicoarsening / line relaxation (point smoother), sslsolver=False.  Golden = the reference itself run in
    the build container (tests/golden/make_golden.py --big solves32; SURVEY App. G: 6 cycles, CONVERGED)."""
    g = load_golden("solves_32.npz")
    h = g['h']
    mesh = oracle.Mesh([h, h, h], (-800., -800., -800.))
    vol = mesh.cell_volumes
    eta = np.asfortranarray(g['smu0'] * vol / 1.0)
    model = oracle.VModel(eta, eta, eta, np.asfortranarray(vol), case=0)
    e, info = oracle.solve(mesh, model, g['sfield'].copy(), cycle='F')
    assert info['exit'] == 0 and info['it_mg'] == len(g['error_at_cycle']) - 1 == 6
    np.testing.assert_allclose(info['error_at_cycle'], g['error_at_cycle'], rtol=1e-7)
    assert relerr(e, g['efield']) < 1e-10
    # known answers of SURVEY App. G
    assert abs(np.linalg.norm(g['efield']) / 2.5774151107694893e-06 - 1) < 1e-12


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_level0_cycmax_is_fixed_on_entry(oracle, tag):
    """solver.multigrid evaluates cycmax of level 0 once, on entry, with the sc_dir of that moment (reference
    emg3d/solver.py:480-485).  On small odd grids the first direction of semicoarsening=True has clevel 0; the children
    of all later F-cycles then get new_cycmax = 1.  Golden = the reference's per-cycle norms and field
    (tests/golden/solves_entry.npz)."""
    g = load_golden("solves_entry.npz")
    mesh = oracle.Mesh([g[f'{tag}_hx'], g[f'{tag}_hy'], g[f'{tag}_hz']], g[f'{tag}_origin'])
    import emg3d_amd as em          # host-side containers only (source vector, eta, zeta)
    grid = em.TensorMesh([g[f'{tag}_hx'], g[f'{tag}_hy'], g[f'{tag}_hz']], origin=g[f'{tag}_origin'])
    rho = g[f'{tag}_rho']
    model = em.Model(grid, rho, property_z=2 * rho)
    sfield = em.get_source_field(grid, g[f'{tag}_src'], float(g[f'{tag}_freq']))
    vm = em.VolumeModel(grid, model, sfield)
    e, info = oracle.solve(mesh, oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield), cycle='F',
                           semicoarsening=True, linerelaxation=int(g[f'{tag}_lr']), nu_init=0, nu_pre=2, nu_coarse=2,
                           nu_post=2, maxit=3, tol=1e-14)
    assert_norms_close(info['error_at_cycle'], g[f'{tag}_error_at_cycle'])
    assert relerr(e, g[f'{tag}_efield']) < 1e-10


@pytest.mark.parametrize("tag", ["f", "s"])
def test_epsilon_r_fixture(oracle, tag):
    """Model with epsilon_r and mu_r (tests/golden/solves_eps.npz, generated by the reference: frequency and Laplace domain):
    the oracle's cycle on the reference's eta / zeta arrays reproduces the reference's F-cycle solve -- and those arrays are
    what emg3d_amd.VolumeModel forms (eta = s mu_0 V (sigma - s eps_0 eps_r), emg3d/models.py:631-647)."""
    import emg3d_amd as em
    g = load_golden("solves_eps.npz")
    mesh = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'], mu_r=g['mu_r'], epsilon_r=g['eps_r'])
    sfield = em.get_source_field(grid, g['src'], float(g[f'{tag}_freq']))
    vm = em.VolumeModel(grid, model, sfield)
    for c in 'xyz':
        np.testing.assert_array_equal(np.asarray(getattr(vm, f'eta_{c}')), g[f'{tag}_eta_{c}'])
    np.testing.assert_array_equal(np.asarray(vm.zeta), g[f'{tag}_zeta'])
    e, info = oracle.solve(mesh, oracle.VModel(g[f'{tag}_eta_x'], g[f'{tag}_eta_y'], g[f'{tag}_eta_z'], g[f'{tag}_zeta']),
                           np.array(sfield), cycle='F', semicoarsening=True, linerelaxation=True)
    assert info['it_mg'] == int(g[f'{tag}_it'])
    assert_norms_close(info['error_at_cycle'], g[f'{tag}_error_at_cycle'])
    assert relerr(e, g[f'{tag}_efield']) < 1e-9
