"""GPU parity, cycle level: ``emg3d_amd.solve`` (device-resident multigrid
through the C-ABI handle) against the reference's golden fields / per-cycle
error traces (lexicographic ordering) and against the CPU oracle (both
orderings)."""
import numpy as np
import pytest

from conftest import assert_norms_close, load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def em():
    import emg3d_amd
    return emg3d_amd


# Parity with the reference in its own (lexicographic) sweep order.  Per-cycle residual norms
# (conftest.assert_norms_close): every cycle whose residual is still > 1e-5 of the source norm agrees to the
# north star's 1e-10 RELATIVE TO ITSELF; the later ones, which are 1e-7 of the first, to NORM_RTOL relative to
# themselves (2e-9 there is 2e-16 of the source norm: the cancellation floor of r = s - A e).  Fields:
# relative max-norm over all edges.
NORM_RTOL = 2e-9
FIELD_TOL = 1e-11


def _reg(em, g, prefix):
    grid = em.TensorMesh([g[f'{prefix}_hx'], g[f'{prefix}_hy'], g[f'{prefix}_hz']], origin=g[f'{prefix}_origin'])
    model = em.Model(grid, g[f'{prefix}_property_x'], g[f'{prefix}_property_y'], g[f'{prefix}_property_z'])
    sfield = em.SourceField(grid, g[f'{prefix}_sfield'].copy(), freq=float(g[f'{prefix}_freq']))
    return grid, model, sfield


@pytest.mark.parametrize("key,kw", [('F', {}), ('W', {'cycle': 'W'}), ('V', {'cycle': 'V'}),
                                    ('bic', {'sslsolver': True})])
def test_regression_res(em, key, kw):
    """reference tests/test_solver.py:28-82 (F, W, V cycles, BiCGSTAB)."""
    g = load_golden("regression.npz")
    grid, model, sfield = _reg(em, g, 'res')
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    assert info['exit'] == 0
    assert info['it_mg'] == g[f'res_{key}_it'][0] and info['it_ssl'] == g[f'res_{key}_it'][1]
    # measured (tools/parity_probe.py): per-cycle norms within 7e-11, fields within 3e-15 of the reference
    assert_norms_close(info['error_at_cycle'], g[f'res_{key}_error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, g[f'res_{key}_here']) < FIELD_TOL      # reference run in the build container
    assert relerr(e, g[f'res_{key}_golden']) < 5e-9    # reference's 2020 golden (other CODATA mu_0)


def test_regression_reg2(em):
    """reference tests/test_solver.py:162-186: sc=123, lr=456, nu_init=2, maxit=4 +
    the 2x2-iterations == 4-iterations warm-start identity."""
    g = load_golden("regression.npz")
    grid, model, sfield = _reg(em, g, 'reg_2')
    kw = {k: g[f'reg_2_inp_{k}'].item() for k in ('semicoarsening', 'linerelaxation', 'tol', 'maxit',
                                                  'nu_init', 'nu_pre', 'nu_coarse', 'nu_post', 'clevel')}
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    assert_norms_close(info['error_at_cycle'], g['reg_2_error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, g['reg_2_here']) < FIELD_TOL
    assert relerr(e, g['reg_2_golden']) < 5e-9
    # warm start: two runs of 1 cycle == one run of 2 cycles
    kw2 = dict(kw, maxit=1, tol=1e-30, nu_init=0, semicoarsening=1, linerelaxation=4)
    e1 = em.solve(grid, model, sfield, ordering='lex', **kw2)
    em.solve(grid, model, sfield, efield=e1, ordering='lex', **kw2)
    e2 = em.solve(grid, model, sfield, ordering='lex', **dict(kw2, maxit=2))
    assert relerr(e1, e2) < 1e-12


@pytest.mark.parametrize("key,kw", [('F', {}), ('bic', {'sslsolver': True})])
def test_regression_lap(em, key, kw):
    """Laplace domain (float64 kernels), reference tests/test_solver.py:267-294."""
    g = load_golden("regression.npz")
    grid, model, sfield = _reg(em, g, 'lap')
    assert sfield.dtype == np.float64
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    assert e.dtype == np.float64
    assert_norms_close(info['error_at_cycle'], g[f'lap_{key}_error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, g[f'lap_{key}_here']) < FIELD_TOL
    assert relerr(e, g[f'lap_{key}_golden']) < 1e-8


def _s16(em, g):
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = em.get_source_field(grid, g['src'], float(g['freq']))
    assert relerr(sfield, g['sfield']) < 1e-14
    return grid, model, sfield


@pytest.mark.parametrize("name,kw", [
    ('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)),
    ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True)),
    ('W_sclr', dict(cycle='W', semicoarsening=True, linerelaxation=True)),
    ('F_plain', dict(cycle='F', maxit=5)),
    ('bic_sclr', dict(sslsolver=True, semicoarsening=True, linerelaxation=True)),
])
def test_solves_16_lex_vs_reference(em, name, kw):
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    assert info['it_mg'] == g[f'{name}_it'][0] and info['it_ssl'] == g[f'{name}_it'][1]
    assert info['exit'] == int(g[f'{name}_exit'])
    assert_norms_close(info['error_at_cycle'], g[f'{name}_error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, g[f'{name}_efield']) < FIELD_TOL


@pytest.mark.parametrize("cycle", ['F', 'V'])
def test_solves_16_colour_vs_oracle(em, oracle, cycle):
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='colour', cycle=cycle,
                       semicoarsening=True, linerelaxation=True)
    vm = em.VolumeModel(grid, model, sfield)
    oe, oinfo = oracle.solve(oracle.Mesh(grid.h, grid.origin),
                             oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield),
                             cycle=cycle, semicoarsening=True, linerelaxation=True, order=1)
    assert info['it_mg'] == oinfo['it_mg'] and info['exit'] == 0
    assert_norms_close(info['error_at_cycle'], oinfo['error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, oe) < FIELD_TOL
    # and the coloured solve agrees with the reference's field to the solver tolerance
    assert relerr(e, g[f'{cycle}_sclr_efield']) < 1e-5


@pytest.mark.parametrize("name,kw", [
    ('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)),
    ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True)),
    ('F_plain', dict(cycle='F', maxit=5)),
    ('bic_sclr', dict(sslsolver=True, semicoarsening=True, linerelaxation=True)),
])
def test_solves_16_colour_vs_reference_arithmetic(em, name, kw):
    """The TIMED ordering at cycle level against reference arithmetic: `solves_16_colour.npz` is the reference's own `solver.solve`
    with its smoothing calls replaced by the sub-grid replay of the device's colour schedule (make_golden.py::colour_solve_fixture).
    Counts and exit status exact; per-cycle norms to the north star's 1e-10 while the residual is above 1e-5 of the source norm
    (conftest.assert_norms_close: the same bar the lexicographic mode meets against the reference's goldens); field <= FIELD_TOL."""
    g = load_golden("solves_16.npz")
    c = load_golden("solves_16_colour.npz")
    grid, model, sfield = _s16(em, g)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='colour', **kw)
    assert info['it_mg'] == c[f'{name}_it'][0] and info['it_ssl'] == c[f'{name}_it'][1] and info['exit'] == int(c[f'{name}_exit'])
    assert_norms_close(info['error_at_cycle'], c[f'{name}_error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, c[f'{name}_efield']) < FIELD_TOL


def test_solves_16_colour_laplace_vs_reference_arithmetic(em):
    """... and in the Laplace domain (s = 2: the float64 kernels of the colour ordering at cycle level)."""
    g = load_golden("solves_16.npz")
    c = load_golden("solves_16_colour.npz")
    grid, model, _ = _s16(em, g)
    sfield = em.get_source_field(grid, list(g['src']), -2.0)
    assert relerr(sfield, c['lap_sfield']) < 1e-14
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='colour', cycle='F', semicoarsening=True,
                       linerelaxation=True)
    assert np.asarray(e).dtype == np.float64
    assert info['it_mg'] == c['lap_F_sclr_it'][0] and info['exit'] == int(c['lap_F_sclr_exit'])
    # (measured: 5.6e-10 on the cycle whose residual is 2.5e-5 of the source norm -- the oracle's twin is at 3.9e-11 there; both grow by
    # ~10 x per cycle as the residual falls: rounding differences of the line solves relative to a shrinking error.  The strict
    # 1e-10 bar therefore ends at 1e-4 of the source norm here, as for the two-sided variants in test_gpu_variants.py.)
    assert_norms_close(info['error_at_cycle'], c['lap_F_sclr_error_at_cycle'], rtol=NORM_RTOL, strict_above=1e-4)
    assert relerr(e, c['lap_F_sclr_efield']) < FIELD_TOL


def test_32cube_cycle_vs_oracle(em, oracle):
    """32^3 stretched tri-axial, 2 F-cycles sc+lr: per-cycle norms, both orderings."""
    h = em.meshes.stretched_widths(16, 8, 100., 1.3)
    grid = em.TensorMesh([h, h, h], origin=(-h.sum() / 2,) * 3)
    rng = np.random.default_rng(1234)
    rho = 10 ** rng.uniform(-0.5, 1.5, grid.nC)
    model = em.Model(grid, rho, 2 * rho, 3 * rho)
    sfield = em.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0)
    vm = em.VolumeModel(grid, model, sfield)
    om = oracle.Mesh(grid.h, grid.origin)
    ov = oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)
    for ordering, order in (('lex', 0), ('colour', 1)):
        e, info = em.solve(grid, model, sfield, return_info=True, ordering=ordering, cycle='F',
                           semicoarsening=True, linerelaxation=True, maxit=2, verb=0)
        oe, oinfo = oracle.solve(om, ov, np.array(sfield), cycle='F', semicoarsening=True,
                                 linerelaxation=True, maxit=2, order=order)
        np.testing.assert_allclose(info['error_at_cycle'], oinfo['error_at_cycle'], rtol=1e-7)
        assert relerr(e, oe) < 1e-10
    # SURVEY App. G trace of the reference for this configuration (first two cycles)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', cycle='F',
                       semicoarsening=True, linerelaxation=True, maxit=2, verb=0)
    np.testing.assert_allclose(info['error_at_cycle'], [5.57230067e-06, 1.24904373e-07, 1.42140420e-08],
                               rtol=1e-7)


def test_termination_paths(em):
    """Zero source, already-converged efield, maxit (reference tests/test_solver.py:113-159)."""
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    zero = em.SourceField(grid, freq=1.0)
    e, info = em.solve(grid, model, zero, return_info=True)
    assert info['exit'] == 0 and not np.asarray(e).any()
    e, info = em.solve(grid, model, sfield, return_info=True, semicoarsening=True, linerelaxation=True)
    info2 = em.solve(grid, model, sfield, efield=e, return_info=True, semicoarsening=True, linerelaxation=True)
    assert info2['it_mg'] == 0 and info2['exit'] == 0
    info3 = em.solve(grid, model, sfield, return_info=True, maxit=1, verb=0)[1]
    assert info3['exit'] == 1 and info3['exit_message'].startswith("MAX. ITERATION")
    with pytest.raises(ValueError):
        em.solve(grid, model, sfield, cycle='X')
    with pytest.raises(ValueError):
        em.solve(grid, model, sfield, efield=em.Field(grid, dtype=np.float64))


@pytest.mark.parametrize("env", [{}, {"EMG3D_SPLIT_MIN_CELLS": "500"}, {"EMG3D_QPL": "0"}])
def test_handle_reuse_with_new_source(em, monkeypatch, request, env):
    """One device handle, two different sources, cycles starting at different
    (sc_dir, lr_dir) pairs (the preconditioner use case): the replayed cycle
    graphs must see the new source in every working copy (transposed, parity-split)."""
    from emg3d_amd.solver import DeviceMG, MGParameters
    if env:                 # tuning variables exist in the lab build only
        request.getfixturevalue("lab")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    vm = em.VolumeModel(grid, model, sfield)
    s2 = em.get_source_field(grid, [50., -30., 20., 110., -20.], float(g['freq']))
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)

    def run(dev, s, keys):
        dev.set_sfield(s)
        dev.set_efield(None)
        norms = [dev.cycle(sc, lr) for sc, lr in keys]
        return np.array(norms), dev.get_efield()

    keys1 = [(1, 4), (2, 5), (3, 6), (1, 4)]
    keys2 = [(3, 6), (2, 5), (1, 4)]          # starts with a pair whose graph was captured late
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var)
        run(dev, sfield, keys1)
        n2, e2 = run(dev, s2, keys2)
    with DeviceMG(grid, vm, np.complex128) as fresh:
        fresh.set_params(var)
        n2f, e2f = run(fresh, s2, keys2)
    np.testing.assert_allclose(n2, n2f, rtol=1e-9)
    assert relerr(e2, e2f) < 1e-10


def test_solve_model_paths_agree_bitwise(em):
    """solve() forms eta on the device from the real factor (models.eta_factored); a handle built from
    the host-side VolumeModel must give the same cycles bit for bit; epsilon_r takes the VolumeModel path."""
    from emg3d_amd.solver import DeviceMG
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    kw = dict(cycle='F', semicoarsening=True, linerelaxation=True, return_info=True)
    e1, i1 = em.solve(grid, model, sfield, **kw)
    vm = em.VolumeModel(grid, model, sfield)
    with DeviceMG(grid, vm, np.complex128) as dev:
        e2, i2 = em.solve(grid, model, sfield, handle=dev, **kw)
    assert np.array_equal(np.array(e1), np.array(e2))
    assert np.array_equal(i1['error_at_cycle'], i2['error_at_cycle'])
    m3 = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'], epsilon_r=np.ones(grid.nC))
    e3, i3 = em.solve(grid, m3, sfield, **kw)
    assert i3['exit'] == 0 and relerr(e3, e1) < 1e-6     # displacement currents are negligible at 1 Hz
    # epsilon_r large enough to matter: device-formed eta (emg3d_mg_create_vse) against the host VolumeModel, bit for bit
    rng = np.random.default_rng(3)
    m4 = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'], epsilon_r=10 ** rng.uniform(6., 8., grid.nC))
    e4, i4 = em.solve(grid, m4, sfield, **kw)
    with DeviceMG(grid, em.VolumeModel(grid, m4, sfield), np.complex128) as dev:
        e5, i5 = em.solve(grid, m4, sfield, handle=dev, **kw)
    assert np.array_equal(np.array(e4), np.array(e5)) and np.array_equal(i4['error_at_cycle'], i5['error_at_cycle'])
    assert relerr(e4, e1) > 1e-4


@pytest.mark.parametrize("tag", ["f", "s"])
def test_epsilon_r_against_the_reference(em, tag):
    """Model with epsilon_r and mu_r (tests/golden/solves_eps.npz: the reference's VolumeModel arrays and F-cycle solves in
    the frequency and in the Laplace domain, reference emg3d/models.py:631-647): the device forms eta from sigma, V and
    eps_r (emg3d_mg_create_vse); solve() in the reference's order reproduces the reference's cycles; a handle re-targeted
    from another frequency (emg3d_mg_set_smu0_eps) is bit for bit a fresh one."""
    from emg3d_amd import models
    from emg3d_amd.solver import DeviceMG
    g = load_golden("solves_eps.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'], mu_r=g['mu_r'], epsilon_r=g['eps_r'])
    freq = float(g[f'{tag}_freq'])
    sfield = em.get_source_field(grid, g['src'], freq)
    vm = em.VolumeModel(grid, model, sfield)
    for c in 'xyz':         # the host restatement of VolumeModel against the reference's arrays
        np.testing.assert_array_equal(np.asarray(getattr(vm, f'eta_{c}')).ravel(order='F'), g[f'{tag}_eta_{c}'].ravel(order='F'))
    e, info = em.solve(grid, model, sfield, cycle='F', semicoarsening=True, linerelaxation=True, return_info=True,
                       ordering='lex', verb=0)
    assert info['it_mg'] == int(g[f'{tag}_it']) and info['exit'] == int(g[f'{tag}_exit'])
    assert_norms_close(info['error_at_cycle'], g[f'{tag}_error_at_cycle'])
    assert relerr(e, g[f'{tag}_efield']) < 1e-9
    # device-formed eta == host VolumeModel: same cycles bit for bit
    with DeviceMG(grid, vm, sfield.dtype) as dev:
        e2, info2 = em.solve(grid, model, sfield, handle=dev, cycle='F', semicoarsening=True, linerelaxation=True,
                             return_info=True, ordering='lex', verb=0)
    assert np.array_equal(np.array(e), np.array(e2)) and np.array_equal(info['error_at_cycle'], info2['error_at_cycle'])
    # a handle that comes from another frequency
    parts = models.model_parts(grid, model, raw=True)
    other = em.fields.FrequencySpec(3 * freq)
    with DeviceMG.from_model(grid, parts, other) as dev:
        dev.set_smu0(sfield.smu0, sval=sfield.sval)
        e3, info3 = em.solve(grid, None, sfield, handle=dev, cycle='F', semicoarsening=True, linerelaxation=True,
                             return_info=True, ordering='lex', verb=0)
        with pytest.raises(TypeError, match="sval"):
            dev.set_smu0(sfield.smu0)
    assert np.array_equal(np.array(e), np.array(e3)) and np.array_equal(info['error_at_cycle'], info3['error_at_cycle'])


@pytest.mark.parametrize("graph", ["1", "0"])
def test_prepare_is_setup_only(em, monkeypatch, graph):
    """emg3d_mg_prepare builds hierarchy / factors / launch graphs but runs no cycle: fields untouched,
    idempotent, and the cycles that follow are those of an unprepared handle."""
    from emg3d_amd.solver import DeviceMG, MGParameters
    monkeypatch.setenv("EMG3D_GRAPH", graph)
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    keys = [(1, 4), (2, 5), (3, 6)]
    e0 = np.asarray(sfield) * (0.5 - 0.25j)
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(e0)
        for sc, lr in keys + keys:
            dev.prepare(sc, lr)
        assert np.array_equal(dev.get_efield(), e0)
        n1 = [dev.cycle(sc, lr) for sc, lr in keys]
        e1 = dev.get_efield()
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(e0)
        n2 = [dev.cycle(sc, lr) for sc, lr in keys]
        e2 = dev.get_efield()
    np.testing.assert_allclose(n1, n2, rtol=1e-12)
    assert relerr(e1, e2) < 1e-13


@pytest.mark.parametrize("vnC,kw", [
    ((48, 24, 20), dict(cycle='F', semicoarsening=True, linerelaxation=True)),
    ((20, 40, 12), dict(cycle='W', semicoarsening=2, linerelaxation=6)),
    ((24, 12, 36), dict(cycle='V', semicoarsening=13, linerelaxation=47, nu_init=1, nu_pre=1, nu_post=3)),
    ((16, 16, 16), dict(cycle='F', semicoarsening=0, linerelaxation=0)),
    ((12, 20, 28), dict(cycle='F', semicoarsening=True, linerelaxation=7, clevel=2)),
])
@pytest.mark.parametrize("ordering", ["lex", "colour"])
def test_ragged_grids_and_parameter_combinations(em, oracle, vnC, kw, ordering):
    """Non-cubic, non-power-of-two grids (3*2^k, 5*2^k, ...) and the
    MGParameters combinations of reference tests/test_solver.py:499-574."""
    rng = np.random.default_rng(sum(vnC))
    h = [rng.uniform(30, 90, n) for n in vnC]
    grid = em.TensorMesh(h, origin=[-hh.sum() / 2 for hh in h])
    rho = 10 ** rng.uniform(-0.3, 1.0, grid.nC)
    model = em.Model(grid, rho, 1.5 * rho, 2.5 * rho, mu_r=rng.uniform(0.9, 1.2, grid.nC))
    sfield = em.get_source_field(grid, [10., -5., 7., 25., 15.], 2.0)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering=ordering, maxit=4, tol=1e-12, verb=0, **kw)
    vm = em.VolumeModel(grid, model, sfield)
    oe, oinfo = oracle.solve(oracle.Mesh(grid.h, grid.origin),
                             oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield),
                             maxit=4, tol=1e-12, order=0 if ordering == 'lex' else 1, **kw)
    assert info['it_mg'] == oinfo['it_mg'] == 4
    assert_norms_close(info['error_at_cycle'], oinfo['error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, oe) < FIELD_TOL


@pytest.mark.parametrize("graph", ["1", "0"])
def test_cycles_reports_every_norm(em, monkeypatch, graph):
    """emg3d_mg_cycles (several cycles enqueued back to back, what bench.py times) returns the residual norm
    of EVERY cycle, identical to cycle-by-cycle calls."""
    from emg3d_amd.solver import DeviceMG, MGParameters
    monkeypatch.setenv("EMG3D_GRAPH", graph)
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    sc, lr = [1, 2, 3], [4, 5, 6]
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None)
        one = [dev.cycle(sc[i % 3], lr[i % 3]) for i in range(5)]
        dev.set_efield(None)
        many = dev.cycles(5, sc, lr)
    assert np.array_equal(np.array(one), many)
    assert one[0] > one[-1] > 0


def test_device_block_pool(em):
    """Blocks of a closed handle are reused by the next one (stale contents must not matter: same result
    bit for bit) and can be handed back to the driver."""
    from emg3d_amd import _lib
    lib = _lib.load()
    g = load_golden("solves_16.npz")
    grid, model, sfield = _s16(em, g)
    kw = dict(cycle='F', semicoarsening=True, linerelaxation=True, return_info=True)
    lib.emg3d_hip_release_cached()
    assert lib.emg3d_hip_cached_bytes() == 0
    e1, i1 = em.solve(grid, model, sfield, **kw)
    held = lib.emg3d_hip_cached_bytes()
    assert held > 0
    e2, i2 = em.solve(grid, model, sfield, **kw)        # runs entirely on recycled blocks
    assert lib.emg3d_hip_cached_bytes() == held
    assert np.array_equal(np.asarray(e1), np.asarray(e2)) and np.array_equal(i1['error_at_cycle'], i2['error_at_cycle'])
    # another problem in between scribbles over the parked blocks of the same sizes
    s2 = em.get_source_field(grid, [50., -30., 20., 110., -20.], 3.0)
    em.solve(grid, model, s2, **kw)
    e3, i3 = em.solve(grid, model, sfield, **kw)
    assert np.array_equal(np.asarray(e1), np.asarray(e3))
    assert lib.emg3d_hip_release_cached() >= held and lib.emg3d_hip_cached_bytes() == 0


def test_config_c1_32cubed(em):
    """BASELINE config C1 on the device (the reference runs it on the CPU only): 32^3 homogeneous isotropic
    fullspace, F-cycle with the POINT smoother (no sc / lr), lexicographic order, against the reference's own
    run (tests/golden/solves_32.npz): same 6 cycles, per-cycle norms, field."""
    g = load_golden("solves_32.npz")
    h = g['h']
    grid = em.TensorMesh([h, h, h], origin=(-800., -800., -800.))
    model = em.Model(grid, 1.)
    sfield = em.get_source_field(grid, [0, 0, 0, 30, 10], 1.0)
    assert relerr(np.array(sfield), g['sfield']) < 1e-14
    e, info = em.solve(grid, model, sfield, cycle='F', return_info=True, ordering='lex', verb=0)
    assert info['exit'] == 0 and info['it_mg'] == 6
    assert_norms_close(info['error_at_cycle'], g['error_at_cycle'], rtol=NORM_RTOL)
    assert relerr(e, g['efield']) < FIELD_TOL


@pytest.mark.parametrize("ordering", ["lex", "colour"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_level0_cycmax_is_fixed_on_entry(oracle, tag, ordering):
    """Found by tests/tools/fuzz_parity.py: solver.multigrid fixes level 0's cycmax on entry from the FIRST sc_dir
    (reference emg3d/solver.py:480-485; emg3d_mg_begin).  8 x 3 x 3 / 48 x 5 x 5 / 12 x 3 x 6 with semicoarsening=True:
    direction 1 has clevel 0, the later F-cycles visit the coarse levels once.  lex: against the reference's own
    norms and field; colour: against the oracle."""
    import emg3d_amd as em
    g = load_golden("solves_entry.npz")
    grid = em.TensorMesh([g[f'{tag}_hx'], g[f'{tag}_hy'], g[f'{tag}_hz']], origin=g[f'{tag}_origin'])
    rho = g[f'{tag}_rho']
    model = em.Model(grid, rho, property_z=2 * rho)
    sfield = em.get_source_field(grid, g[f'{tag}_src'], float(g[f'{tag}_freq']))
    opts = dict(cycle='F', semicoarsening=True, linerelaxation=int(g[f'{tag}_lr']), nu_init=0, nu_pre=2, nu_coarse=2,
                nu_post=2, maxit=3, tol=1e-14)
    e, info = em.solve(grid, model, sfield, return_info=True, verb=0, ordering=ordering, **opts)
    if ordering == 'lex':
        ref_e, ref_n = g[f'{tag}_efield'], g[f'{tag}_error_at_cycle']
    else:
        vm = em.VolumeModel(grid, model, sfield)
        ref_e, oinfo = oracle.solve(oracle.Mesh(grid.h, grid.origin), oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta),
                                    np.array(sfield), order=1, **opts)
        ref_n = oinfo['error_at_cycle']
    # (the wrong cycmax shows as 1e-4 ... 3e-2 in the norms of cycles 2, 3; these tiny, badly conditioned Laplace-domain
    # systems reach rounding level within three cycles, hence the loose bar on the late norms: the device skips the
    # repeated colour at a sweep's turn-around, the oracle does not -- 3e-8 on a norm 1.2e-3 below the first in case b)
    assert_norms_close(info['error_at_cycle'], ref_n, rtol=1e-5, strict_rtol=1e-8, strict_above=1e-2)
    assert relerr(e, ref_e) < 1e-9


def test_solve_special_paths():
    """solve() paths without a cycle, as the reference behaves (emg3d/solver.py:343-369; checked against a run of the
    reference): a provided field that is already good enough (only the info dict comes back, nothing is done), a zero
    source (zero field, abs_error keeps its initial 1.0, relative errors NaN), a zero source with a provided field (the
    field is left alone, its residual is reported)."""
    import emg3d_amd as em
    h = np.ones(8) * 50.
    grid = em.TensorMesh([h, h, h], origin=(-200., -200., -200.))
    model = em.Model(grid, 1.)
    sfield = em.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0)
    e, info = em.solve(grid, model, sfield, return_info=True, verb=0, ordering='lex')
    assert info['it_mg'] == 6 and info['exit'] == 0
    e2 = e.copy()
    out = em.solve(grid, model, sfield, efield=e2, return_info=True, verb=0, ordering='lex')
    assert isinstance(out, dict) and out['it_mg'] == 0 and out['exit'] == 0 and out['exit_message'] == 'CONVERGED'
    assert len(out['error_at_cycle']) == 1 and np.array_equal(np.array(e2), np.array(e))
    assert abs(out['abs_error'] / info['abs_error'] - 1) < 1e-6 and out['rel_error'] < 1e-6
    assert em.solve(grid, model, sfield, efield=e2, verb=0) is None                # reference: nothing to return
    zero = em.SourceField(grid, freq=1.0)
    ez, iz = em.solve(grid, model, zero, return_info=True, verb=0)
    assert not np.array(ez).any() and iz['it_mg'] == 0 and iz['exit'] == 0 and iz['exit_message'] == 'CONVERGED'
    assert iz['abs_error'] == 1.0 and np.isnan(iz['rel_error']) and np.isnan(iz['ref_error'])
    assert list(iz['error_at_cycle']) == [0.0]
    e3 = e.copy()
    i3 = em.solve(grid, model, zero, efield=e3, return_info=True, verb=0)
    assert isinstance(i3, dict) and i3['it_mg'] == 0 and i3['exit_message'] == 'CONVERGED' and np.isnan(i3['rel_error'])
    assert np.array_equal(np.array(e3), np.array(e))                               # the provided field is left alone
    assert abs(i3['abs_error'] / 5.561533017428965e-06 - 1) < 1e-3                # ||A e|| of that field (reference run)


def test_prepare_ahead_changes_nothing(em):
    """emg3d_mg_cycle_next (the next cycle's hierarchy / factorisations / launch graph are set up on the host while the
    device runs the current cycle) against the plain order: fields and per-cycle norms bit for bit, single and batched,
    rotating and fixed directions, maxit cut-offs."""
    from emg3d_amd import solver
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = em.get_source_field(grid, g['src'], float(g['freq']))
    srcs = [list(g['src']), [30., -20., 10., 45., -20.]]
    for kw in (dict(cycle='F', semicoarsening=True, linerelaxation=True), dict(cycle='V', semicoarsening=132, linerelaxation=6),
               dict(cycle='W', semicoarsening=2, linerelaxation=True, maxit=2), dict(cycle='F', semicoarsening=True, maxit=1)):
        got = []
        for ahead in (False, True):
            solver.PREPARE_AHEAD = ahead
            try:
                e, info = em.solve(grid, model, sfield, return_info=True, verb=0, **kw)
                eb, infos = solver.solve_sources(grid, model, srcs, float(g['freq']), verb=0, **kw)
            finally:
                solver.PREPARE_AHEAD = True
            got.append((np.array(e), np.array(info['error_at_cycle']), info['it_mg'], [np.array(x) for x in eb],
                        [np.array(i['error_at_cycle']) for i in infos]))
        a, b = got
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2]
        for x, y in zip(a[3] + a[4], b[3] + b[4]):
            np.testing.assert_array_equal(x, y)
