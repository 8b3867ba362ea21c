"""GPU: device-resident BiCGSTAB -- the vector primitives of the C ABI against
numpy, and the device iteration against (i) the reference's `res>bicresult` / `lap>bicresult`
goldens (tests/test_gpu_solver.py runs them through the default device path), (ii) the
host-SciPy path of the same module (what the reference calls, solver.py:717-719)."""
import numpy as np
import pytest

from conftest import assert_norms_close, load_golden, relerr

pytestmark = pytest.mark.gpu


def _res(em):
    g = load_golden("regression.npz")
    grid = em.TensorMesh([g['res_hx'], g['res_hy'], g['res_hz']], origin=g['res_origin'])
    model = em.Model(grid, g['res_property_x'], g['res_property_y'], g['res_property_z'])
    sfield = em.SourceField(grid, g['res_sfield'].copy(), freq=float(g['res_freq']))
    return g, grid, model, sfield


@pytest.mark.parametrize("laplace", [False, True])
def test_vector_primitives(laplace):
    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG
    g = load_golden("regression.npz")
    pre = 'lap' if laplace else 'res'
    grid = em.TensorMesh([g[f'{pre}_hx'], g[f'{pre}_hy'], g[f'{pre}_hz']], origin=g[f'{pre}_origin'])
    model = em.Model(grid, g[f'{pre}_property_x'], g[f'{pre}_property_y'], g[f'{pre}_property_z'])
    sfield = em.SourceField(grid, g[f'{pre}_sfield'].copy(), freq=float(g[f'{pre}_freq']))
    vm = em.VolumeModel(grid, model, sfield)
    dev = DeviceMG(grid, vm, sfield.dtype)
    rng = np.random.default_rng(3)

    def rnd():
        a = rng.standard_normal(grid.nE)
        return a if laplace else a + 1j * rng.standard_normal(grid.nE)

    a, b = rnd(), rnd()
    alpha = 0.3 if laplace else 0.3 - 1.7j
    try:
        dev.vec_alloc(4)
        dev.vec_set(0, a)
        dev.vec_set(1, b)
        assert np.array_equal(dev.vec_get(0), a)
        ref = np.vdot(a, b) if not laplace else np.dot(a, b)
        assert abs(dev.vec_dot(0, 1) - ref) <= 1e-13 * abs(ref) + 1e-13 * np.linalg.norm(a) * np.linalg.norm(b)
        assert abs(dev.vec_norm(0) - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
        dev.vec_axpy(0, alpha, 1)
        assert relerr(dev.vec_get(0), a + alpha * b) < 1e-15
        dev.vec_scale(1, alpha)
        assert relerr(dev.vec_get(1), alpha * b) < 1e-15
        dev.vec_copy(2, 0)
        assert np.array_equal(dev.vec_get(2), dev.vec_get(0))
        # dst = A src on device vectors == the host-vector entry point
        x = em.Field(grid, rnd(), freq=float(g[f'{pre}_freq']))
        x.ensure_pec
        dev.vec_set(2, np.asarray(x))
        dev.vec_amatvec(3, 2)
        assert relerr(dev.vec_get(3), dev.amatvec(np.asarray(x))) < 1e-15
        # source / field aliases
        dev.vec_copy(dev.SFIELD, 2)
        dev.vec_copy(1, dev.SFIELD)
        assert np.array_equal(dev.vec_get(1), np.asarray(x))
        # error paths: unknown ids, aliasing y = x
        with pytest.raises(Exception):
            dev.vec_copy(9, 0)
        with pytest.raises(Exception):
            dev.vec_axpy(0, 1.0, 0)
    finally:
        dev.close()


@pytest.mark.parametrize("kw", [dict(), dict(semicoarsening=True, linerelaxation=True), dict(cycle=None, maxit=300),
                                dict(ordering='colour', semicoarsening=True, linerelaxation=True)])
def test_device_bicgstab_equals_host_scipy(monkeypatch, kw):
    """Same iteration, vectors on the device vs. SciPy's bicgstab on host vectors."""
    import emg3d_amd as em
    from emg3d_amd import solver
    g, grid, model, sfield = _res(em)
    kw = dict(dict(ordering='lex'), **kw)
    e_dev, i_dev = em.solve(grid, model, sfield, return_info=True, sslsolver='bicgstab', **kw)
    monkeypatch.setattr(solver, 'DEVICE_KRYLOV', False)
    e_host, i_host = em.solve(grid, model, sfield, return_info=True, sslsolver='bicgstab', **kw)
    assert i_dev['exit'] == i_host['exit'] == 0
    if kw.get('cycle', 'F') is None:
        # un-preconditioned BiCGSTAB takes ~90 erratic iterations: the summation order of the dot
        # products (device tree vs numpy pairwise) decides the count; both must converge to one field
        assert abs(i_dev['it_ssl'] - i_host['it_ssl']) < 20
        assert relerr(e_dev, e_host) < 1e-4
        return
    assert i_dev['it_ssl'] == i_host['it_ssl'] and i_dev['it_mg'] == i_host['it_mg']
    assert_norms_close(i_dev['error_at_cycle'], i_host['error_at_cycle'])
    assert relerr(e_dev, e_host) < 1e-11


def test_device_bicgstab_warm_start_and_maxit():
    """x0 != 0 (r = b - A x0 branch) and the 'not converged' exit (info > 0)."""
    import emg3d_amd as em
    g, grid, model, sfield = _res(em)
    e1, i1 = em.solve(grid, model, sfield, return_info=True, sslsolver='bicgstab', ordering='lex', maxit=1)
    assert i1['exit'] == 1 and 'MAX. ITERATION' in i1['exit_message']
    # efield given: updated in place, only the info dict is returned (reference solver.py:399-405)
    i2 = em.solve(grid, model, sfield, efield=e1, return_info=True, sslsolver='bicgstab', ordering='lex')
    assert i2['exit'] == 0
    assert relerr(e1, g['res_bic_here']) < 1e-5


@pytest.mark.parametrize("kw", [dict(), dict(semicoarsening=True, linerelaxation=True),
                                dict(ordering='colour', semicoarsening=True, linerelaxation=True), dict(maxit=2)])
def test_device_cgs_equals_host_scipy(monkeypatch, kw):
    """cgs with every vector on the device vs. SciPy's cgs on host vectors (device operator and preconditioner):
    same iteration and cycle counts, residual histories and fields."""
    import emg3d_amd as em
    from emg3d_amd import solver
    g, grid, model, sfield = _res(em)
    kw = dict(dict(ordering='lex'), **kw)
    e_dev, i_dev = em.solve(grid, model, sfield, return_info=True, sslsolver='cgs', **kw)
    monkeypatch.setattr(solver, 'DEVICE_KRYLOV', False)
    e_host, i_host = em.solve(grid, model, sfield, return_info=True, sslsolver='cgs', **kw)
    assert i_dev['exit'] == i_host['exit'] and i_dev['exit_message'] == i_host['exit_message']
    assert i_dev['it_ssl'] == i_host['it_ssl'] and i_dev['it_mg'] == i_host['it_mg']
    assert_norms_close(i_dev['error_at_cycle'], i_host['error_at_cycle'])
    assert relerr(e_dev, e_host) < 1e-11


@pytest.mark.parametrize("kw", [dict(), dict(semicoarsening=True, linerelaxation=True),
                                dict(ordering='colour', semicoarsening=True, linerelaxation=True), dict(maxit=2),
                                dict(cycle=None)])
def test_device_gcrotmk_equals_host_scipy(monkeypatch, kw):
    """gcrotmk with every vector on the device (SciPy's GCROT(m,k) and its inner FGMRES restated on emg3d_mg_vec_*) vs.
    SciPy's gcrotmk on host vectors (device operator and preconditioner): same iteration and cycle counts, exit, residual
    histories and fields -- preconditioned by F-, V-cycles and not at all (many inner steps, recycled outer vectors)."""
    import emg3d_amd as em
    from emg3d_amd import solver
    g, grid, model, sfield = _res(em)
    kw = dict(dict(ordering='lex'), **kw)
    e_dev, i_dev = em.solve(grid, model, sfield, return_info=True, sslsolver='gcrotmk', **kw)
    monkeypatch.setattr(solver, 'DEVICE_KRYLOV', False)
    e_host, i_host = em.solve(grid, model, sfield, return_info=True, sslsolver='gcrotmk', **kw)
    assert i_dev['exit'] == i_host['exit'] and i_dev['exit_message'] == i_host['exit_message']
    assert i_dev['it_ssl'] == i_host['it_ssl'] and i_dev['it_mg'] == i_host['it_mg']
    assert_norms_close(i_dev['error_at_cycle'], i_host['error_at_cycle'], rtol=1e-6)
    assert relerr(e_dev, e_host) < 1e-8


@pytest.mark.parametrize("name", ['cgs', 'gcrotmk'])
def test_other_krylov_solvers_stay_on_host(oracle, name):
    """cgs / gcrotmk (device resident): same outcome
    as the CPU oracle -- including gcrotmk's DIVERGED exit on this problem with SciPy >= 1.14."""
    import emg3d_amd as em
    g, grid, model, sfield = _res(em)
    e, info = em.solve(grid, model, sfield, return_info=True, sslsolver=name, ordering='lex')
    vm = em.VolumeModel(grid, model, sfield)
    oe, oinfo = oracle.solve(oracle.Mesh(grid.h, grid.origin), oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta),
                             np.array(sfield), sslsolver=name)
    assert info['exit'] == oinfo['exit'] and info['exit_message'] == oinfo['exit_message']
    assert info['it_ssl'] == oinfo['it_ssl'] and info['it_mg'] == oinfo['it_mg']
    if info['exit'] == 0:
        assert relerr(e, oe) < 1e-8


def test_handle_from_sigma_volume_and_frequency_loop():
    """eta = s mu_0 * (sigma V) formed on the device (emg3d_mg_create_sv) and the source from its real
    vector (emg3d_mg_set_sfield_vector): same solve as the host-built VolumeModel / SourceField;
    shard.solve_frequencies == independent solves per frequency."""
    import emg3d_amd as em
    from emg3d_amd import models, shard
    from emg3d_amd.solver import DeviceMG
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    freqs = [float(g['freq']), 2.5]
    opts = dict(cycle='F', semicoarsening=True, linerelaxation=True, ordering='lex')
    res = shard.solve_frequencies(grid, model, g['src'], freqs, **opts)
    for f, (e, info) in zip(freqs, res):
        sfield = em.get_source_field(grid, g['src'], f)
        e0, info0 = em.solve(grid, model, sfield, return_info=True, **opts)
        assert info['it_mg'] == info0['it_mg'] and info['exit'] == 0
        # eta differs by one rounding (smu0*(V*sigma) vs (smu0*V)*sigma): histories agree to ~1e-8
        np.testing.assert_allclose(info['error_at_cycle'], info0['error_at_cycle'], rtol=1e-6)
        assert relerr(e, e0) < 1e-10
    assert relerr(res[0][0], g['F_sclr_efield']) < 1e-11      # the reference's field at the golden frequency
    # the source from its real vector, scaled on the device
    sfield = em.get_source_field(grid, g['src'], freqs[0])
    sv = models.sigma_volume(grid, model)
    with DeviceMG.from_sigma_volume(grid, *sv, smu0=sfield.smu0) as dev:
        dev.set_sfield_vector(sfield.vector, sfield.smu0)
        dev.vec_alloc(1)
        dev.vec_copy(0, dev.SFIELD)
        assert relerr(dev.vec_get(0), np.asarray(sfield)) < 1e-15
    # Laplace domain: real smu0, float64 handle
    sl = em.get_source_field(grid, g['src'], -3.0)
    with DeviceMG.from_sigma_volume(grid, *sv, smu0=sl.smu0) as dev:
        assert dev.dtype == np.float64
        e, info = em.solve(grid, None, sl, handle=dev, return_info=True, **opts)
    e0, info0 = em.solve(grid, model, sl, return_info=True, **opts)
    assert info['it_mg'] == info0['it_mg'] and relerr(e, e0) < 1e-10


def test_concurrent_frequency_solves_are_bitwise_the_sequential_ones():
    """shard.solve_frequencies(concurrent=3): three handles / streams driven by three host threads at the
    same time give exactly the fields and histories of the one-after-the-other run."""
    import emg3d_amd as em
    from emg3d_amd import shard
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    freqs = [float(g['freq']), 2.5, 0.3, 7.0, 0.9]
    opts = dict(cycle='F', semicoarsening=True, linerelaxation=True)
    seq = shard.solve_frequencies(grid, model, g['src'], freqs, **opts)
    par = shard.solve_frequencies(grid, model, g['src'], freqs, concurrent=3, **opts)
    assert len(par) == len(freqs)
    for (e0, i0), (e1, i1) in zip(seq, par):
        assert i0['exit'] == 0 and i1['it_mg'] == i0['it_mg']
        assert np.array_equal(i0['error_at_cycle'], i1['error_at_cycle'])
        assert np.array_equal(np.asarray(e0), np.asarray(e1))
    assert shard.solve_frequencies(grid, model, g['src'], [], concurrent=3) == []


@pytest.mark.parametrize("opts", [dict(cycle='F', semicoarsening=True, linerelaxation=True),
                                  dict(cycle='V', semicoarsening=True, linerelaxation=True, ordering='lex'),
                                  dict(cycle='W', semicoarsening=False, linerelaxation=False),
                                  dict(cycle='F', semicoarsening=2, linerelaxation=4),
                                  dict(cycle='F', semicoarsening=True, linerelaxation=True, sslsolver='bicgstab')])
@pytest.mark.parametrize("iso", [False, True])
def test_handle_reuse_across_frequencies_is_bitwise_a_fresh_handle(opts, iso):
    """emg3d_mg_set_smu0 (shard.solve_frequencies, one after the other: ONE handle per dtype, re-targeted per frequency --
    eta, coarse models, transposed copies, line factorisations recomputed; hierarchy, buffers, launch graphs kept) against
    one fresh handle per frequency (the concurrent path): fields, histories, iteration counts and receiver responses bit
    for bit, over frequency- AND Laplace-domain values in one list, tri-axial and isotropic (aliased eta) models."""
    import emg3d_amd as em
    from emg3d_amd import shard
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b']) if iso else em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    freqs = [float(g['freq']), -2.0, 0.3, 7.0, -0.5, float(g['freq'])]
    rec = (np.array([100., -150.]), np.array([50., 20.]), np.array([-80., 60.]), np.array([0., 30.]), np.array([0., 10.]))
    seq = shard.solve_frequencies(grid, model, g['src'], freqs, rec=rec, **opts)
    par = shard.solve_frequencies(grid, model, g['src'], freqs, rec=rec, concurrent=2, **opts)
    for (e0, i0, r0), (e1, i1, r1) in zip(seq, par):
        assert i1['it_mg'] == i0['it_mg'] and i1['it_ssl'] == i0['it_ssl'] and i1['exit'] == i0['exit']
        assert np.array_equal(i0['error_at_cycle'], i1['error_at_cycle'])
        assert np.array_equal(np.asarray(e0), np.asarray(e1))
        assert np.array_equal(np.asarray(r0), np.asarray(r1))
    # the same frequency again (last entry) after the handle has been elsewhere: the first result, bit for bit
    assert np.array_equal(np.asarray(seq[0][0]), np.asarray(seq[-1][0]))


def test_set_smu0_argument_checks():
    import emg3d_amd as em
    from emg3d_amd import models
    from emg3d_amd.solver import DeviceMG
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'])
    sf = em.SourceField(grid, freq=1.0)
    with DeviceMG(grid, em.VolumeModel(grid, model, sf), sf.dtype) as dev:          # made from eta arrays: no sigma*V
        with pytest.raises(RuntimeError, match="emg3d_mg_set_smu0"):
            dev.set_smu0(sf.smu0)
    with DeviceMG.from_sigma_volume(grid, *models.sigma_volume(grid, model), smu0=sf.smu0) as dev:
        with pytest.raises(ValueError, match="complex"):
            dev.set_smu0(3.0)
        dev.set_smu0(2 * sf.smu0)
    sl = em.SourceField(grid, freq=-1.0)
    with DeviceMG.from_sigma_volume(grid, *models.sigma_volume(grid, model), smu0=sl.smu0) as dev:
        with pytest.raises(ValueError, match="real"):
            dev.set_smu0(1j)


@pytest.mark.parametrize("iso", [False, True])
def test_frequency_and_survey_loops_are_bitwise_plain_solves(iso):
    """Handles made from (sigma, V) (emg3d_mg_create_vs) form eta = (s mu_0 V) sigma on the device with VolumeModel's
    rounding -- at creation and after every emg3d_mg_set_smu0.  Hence shard.solve_frequencies (one re-targeted handle) is
    bit for bit em.solve() per frequency (host-built VolumeModel semantics), and shard.solve_survey (re-targeted batched
    handles) bit for bit solver.solve_sources per frequency; Laplace- and frequency-domain values mixed."""
    import emg3d_amd as em
    from emg3d_amd import shard, solver
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], mu_r=1 + 0 * g['rho_b']) if iso else em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    freqs = [float(g['freq']), 0.3, -2.0, 7.0]
    opts = dict(cycle='F', semicoarsening=True, linerelaxation=True, verb=0)
    res = shard.solve_frequencies(grid, model, g['src'], freqs, **opts)
    for f, (e, info) in zip(freqs, res):
        e0, info0 = em.solve(grid, model, em.get_source_field(grid, g['src'], f), return_info=True, **opts)
        # (the source: solve_frequencies builds it in HBM, em.solve uploads the host-built one -- pinned equal to 1e-14,
        #  not bit for bit; the OPERATOR is what this test is about: same eta => same iteration counts, fields to rounding)
        assert info['it_mg'] == info0['it_mg']
        assert relerr(e, e0) < 1e-12
        e1, info1 = em.solve(grid, model, em.SourceField(grid, freq=f), source=(g['src'], 0), return_info=True, **opts)
        assert np.array_equal(np.asarray(e), np.asarray(e1))                     # device-built source in both: bit for bit
        assert np.array_equal(info['error_at_cycle'], info1['error_at_cycle'])
    srcs = [list(g['src']), [30., -20., 10., 45., -20.], [-50., 40., -30., 10., 80.]]
    rec = (np.array([100., -150.]), np.array([50., 20.]), np.array([-80., 60.]), np.array([0., 30.]), np.array([0., 10.]))
    resp, infos, efs = shard.solve_survey(grid, model, srcs, freqs, rec, batch=2, return_fields=True, **opts)
    for jf, f in enumerate(freqs):
        for i0 in (0, 2):
            chunk = srcs[i0:i0 + 2]
            e, info, r = solver.solve_sources(grid, model, chunk, f, rec=rec, **opts)
            for k in range(len(chunk)):
                assert np.array_equal(np.asarray(efs[i0 + k][jf]), np.asarray(e[k]))
                got = np.asarray(resp[i0 + k, jf])
                assert np.array_equal(got if r.dtype.kind == 'c' else got.real, r[k])
                assert infos[i0 + k][jf]['it_mg'] == info[k]['it_mg']
