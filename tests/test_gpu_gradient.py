"""GPU: adjoint-state gradient pieces against the reference (tests/golden/gradient.npz: the
reference's get_source_field / solve / get_receiver_response / edges2cellaverages composed the way
optimize.gradient and Simulation._get_rfield / _get_bfields compose them) and the oracle."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def _grid(em, g):
    return em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])


def test_edges2cellaverages_vs_reference():
    import emg3d_amd as em
    g = load_golden("gradient.npz")
    grid = _grid(em, g)
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    for tag, freq in (('c', 1.), ('r', -1.)):
        f = em.Field(grid, g[f'e2c_{tag}_in'].copy(), freq=freq)
        ox = np.zeros(grid.vnC, order='F', dtype=f.dtype); oy = ox.copy(); oz = ox.copy()
        em.maps.edges2cellaverages(ex=f.fx, ey=f.fy, ez=f.fz, vol=vol, out_x=ox, out_y=oy, out_z=oz)
        for got, key in ((ox, 'x'), (oy, 'y'), (oz, 'z')):
            assert relerr(got, g[f'e2c_{tag}_{key}']) < 1e-15
        # the reference ADDS into the outputs
        em.maps.edges2cellaverages(ex=f.fx, ey=f.fy, ez=f.fz, vol=vol, out_x=ox, out_y=oy, out_z=oz)
        assert relerr(ox, 2 * g[f'e2c_{tag}_x']) < 1e-15


def test_edges2cellaverages_explicit():
    """reference tests/test_maps.py:439-484: a 2x2x2 mesh with distinct cell widths, one non-zero edge per direction, all
    eight cells checked against volume x (fields on the cell's edges) / 4."""
    import emg3d_amd as em
    x0, x1, y0, y1, z0, z1 = 2, 3, 4, 5, 6, 7
    grid = em.TensorMesh([[x0, x1], [y0, y1], [z0, z1]], origin=(0, 0, 0))
    field = em.Field(grid)
    fx, fy, fz = 1.23 + 9.87j, 2.68 - 5.48j, 1.57 + 7.63j
    field.fx[0, 1, 1] = fx
    field.fy[1, 1, 1] = fy
    field.fz[1, 1, 0] = fz
    gx = np.zeros(grid.vnC, order='F', dtype=complex); gy = gx.copy(); gz = gx.copy()
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    em.maps.edges2cellaverages(field.fx, field.fy, field.fz, vol, gx, gy, gz)
    grad = gx + gy + gz
    want = {(0, 0, 0): x0*y0*z0*(fx+fz)/4, (1, 0, 0): x1*y0*z0*fz/4, (0, 1, 0): x0*y1*z0*(fx+fy+fz)/4,
            (1, 1, 0): x1*y1*z0*(fy+fz)/4, (0, 0, 1): x0*y0*z1*fx/4, (1, 0, 1): 0j, (0, 1, 1): x0*y1*z1*(fx+fy)/4,
            (1, 1, 1): x1*y1*z1*fy/4}
    for ijk, w in want.items():
        assert abs(grad[ijk] - w) <= 1e-14 * max(abs(w), 1), ijk


def test_gradient_vs_reference():
    import emg3d_amd as em
    g = load_golden("gradient.npz")
    grid = _grid(em, g)
    model = em.Model(grid, g['res'])
    rec = tuple(g['rec'])
    phi, grad, info = em.optimize.gradient(grid, model, g['src'], float(g['freq']), rec, g['observed'], g['weights'],
                                           cycle='F', semicoarsening=True, linerelaxation=True, tol=1e-8, verb=0,
                                           ordering='lex')
    # both solves are iterative (tol 1e-8 relative residual): data, misfit and gradient agree to that level
    assert info['forward']['exit'] == 0 and info['backward']['exit'] == 0
    assert relerr(info['synthetic'], g['synthetic']) < 1e-6
    assert abs(phi / float(g['misfit']) - 1) < 1e-5
    assert grad.shape == tuple(grid.vnC)
    assert relerr(grad, g['grad']) < 1e-5
    # the gradient kernel itself, fed with the reference's OWN forward and back-propagated fields: rounding level
    from emg3d_amd.solver import DeviceMG
    from emg3d_amd import models
    sf = em.SourceField(grid, freq=float(g['freq']))
    with DeviceMG.from_sigma_volume(grid, *models.sigma_volume(grid, model), smu0=sf.smu0) as dev:
        dev.vec_alloc(1)
        dev.vec_set(0, g['efield'])
        dev.set_efield(em.Field(grid, g['bfield'].copy(), freq=float(g['freq'])))
        got = dev.gradient(0, sf.smu0).reshape(grid.vnC, order='F')
    assert relerr(got, g['grad']) < 1e-14
    # the residual source built on the device == the reference's residual field
    from oracle import gradient as og
    _, res = og.misfit(g['synthetic'], g['observed'], g['weights'])
    st = og.residual_strengths(res, g['weights'], g['smu0'])
    with DeviceMG.from_sigma_volume(grid, *models.sigma_volume(grid, model), smu0=sf.smu0) as dev:
        for i in range(res.size):
            dev.set_source(g['rec'][:, i], sf.smu0, strength=st[i], accumulate=i > 0)
        assert relerr(dev.vec_get(dev.SFIELD), g['rfield']) < 1e-12


def test_gradient_is_the_derivative_of_the_misfit():
    """Finite-difference check in the reference's own test set-up (tests/test_optimize.py:147-210, scaled down):
    uniform grid, source and receiver well apart, cells between them.  The reference returns
    grid2grid(-grad) as d(misfit)/d(conductivity) (optimize.py:201-203), so -grad is compared with the finite
    difference (measured on the CPU oracle: agreement 1e-5)."""
    import emg3d_amd as em
    hx, hy, hz = np.ones(24) * 100., np.ones(16) * 100., np.ones(16) * 100.
    grid = em.TensorMesh([hx, hy, hz], origin=(0., 0., 0.))
    src = [450., 800., 800., 0., 0.]
    rec = (np.array([1950.]), np.array([800.]), np.array([800.]), np.array([0.]), np.array([0.]))
    kw = dict(cycle='F', tol=1e-11, maxit=80, verb=0)
    sig_true = np.ones(grid.vnC)
    sig_true[9:14, 6:10, 5:9] = 0.01
    e_obs = em.solve(grid, em.Model(grid, sig_true, mapping='Conductivity'), em.get_source_field(grid, src, 1.0), **kw)
    obs = em.get_receiver_response(grid, e_obs, rec)
    w = 1 / (0.05 * np.abs(obs)) ** 2
    sig = np.ones(grid.vnC)
    phi0, grad, info = em.optimize.gradient(grid, em.Model(grid, sig, mapping='Conductivity'), src, 1.0, rec, obs, w, **kw)
    assert phi0 > 0 and info['backward']['exit'] == 0
    for ijk in [(11, 8, 7), (10, 9, 6), (14, 8, 8)]:
        d = 1e-4
        s2 = sig.copy()
        s2[ijk] += d
        phi1, _, _ = em.optimize.gradient(grid, em.Model(grid, s2, mapping='Conductivity'), src, 1.0, rec, obs, w, **kw)
        fd = (phi1 - phi0) / d
        assert abs(fd / -grad[ijk] - 1) < 1e-3, (ijk, fd, -grad[ijk])


def test_gradient_magnetic_receivers_vs_reference():
    """Magnetic receivers (optimize.gradient(..., electric=False)): data = responses of H = get_h_field(E), residual sources
    = magnetic point dipoles (square loops) of strength conj(r) conj(w) / smu0^2 (reference simulations.py:1190-1197);
    fixture composed from the reference's own functions."""
    import emg3d_amd as em
    g = load_golden("gradient.npz")
    grid = _grid(em, g)
    model = em.Model(grid, g['res'])
    rec = tuple(g['rec'])
    phi, grad, info = em.optimize.gradient(grid, model, g['src'], float(g['freq']), rec, g['m_observed'], g['m_weights'],
                                           electric=False, cycle='F', semicoarsening=True, linerelaxation=True, tol=1e-8,
                                           verb=0, ordering='lex')
    assert info['forward']['exit'] == 0 and info['backward']['exit'] == 0
    assert relerr(info['synthetic'], g['m_synthetic']) < 1e-6
    assert abs(phi / float(g['m_misfit']) - 1) < 1e-5
    assert relerr(grad, g['m_grad']) < 1e-5
    # the residual source alone, from the reference's residuals: loops of four dipoles per receiver, negated
    from emg3d_amd.solver import DeviceMG
    from emg3d_amd import models
    sf = em.SourceField(grid, freq=float(g['freq']))
    res = g['m_synthetic'] - g['m_observed']
    with DeviceMG.from_sigma_volume(grid, *models.sigma_volume(grid, model), smu0=sf.smu0) as dev:
        for i in range(res.size):
            st = res[i].conj() * np.conj(g['m_weights'][i]) / sf.smu0 / sf.smu0
            dev.set_source(g['rec'][:, i], sf.smu0, strength=st, accumulate=i > 0, electric=False)
        assert relerr(dev.vec_get(dev.SFIELD), g['m_rfield']) < 1e-12
