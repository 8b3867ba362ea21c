"""Pin the ORACLE's gradient pieces (oracle/gradient.py) against the reference (tests/golden/gradient.npz:
maps.edges2cellaverages on random fields; misfit, residual source and adjoint-state gradient of one
(source, frequency) pair composed from the reference's own functions).  CPU only."""
import numpy as np

from conftest import load_golden, relerr


def _parts(g):
    h = [g['hx'], g['hy'], g['hz']]
    vnC = tuple(a.size for a in h)
    vol = (h[0][:, None, None] * h[1][None, :, None] * h[2][None, None, :])
    nx, ny, nz = vnC
    shp = ((nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz))
    off = np.cumsum([0] + [int(np.prod(s)) for s in shp])
    return h, vnC, vol, shp, off


def test_edges2cellaverages():
    from oracle import gradient as og
    g = load_golden("gradient.npz")
    h, vnC, vol, shp, off = _parts(g)
    for tag in ('c', 'r'):
        f = g[f'e2c_{tag}_in']
        comps = [f[off[c]:off[c + 1]].reshape(shp[c], order='F') for c in range(3)]
        ox, oy, oz = og.edges2cellaverages(*comps, vol)
        for got, key in ((ox, 'x'), (oy, 'y'), (oz, 'z')):
            assert np.array_equal(got, g[f'e2c_{tag}_{key}'])          # same statements, same order: bit-identical


def test_edges2cellaverages_explicit():
    """reference tests/test_maps.py:439-484 (2x2x2 mesh, one non-zero edge per direction, eight cells by hand)."""
    from oracle import gradient as og
    x0, x1, y0, y1, z0, z1 = 2., 3., 4., 5., 6., 7.
    vol = np.array([x0, x1])[:, None, None] * np.array([y0, y1])[None, :, None] * np.array([z0, z1])[None, None, :]
    ex = np.zeros((2, 3, 3), complex); ey = np.zeros((3, 2, 3), complex); ez = np.zeros((3, 3, 2), complex)
    fx, fy, fz = 1.23 + 9.87j, 2.68 - 5.48j, 1.57 + 7.63j
    ex[0, 1, 1] = fx; ey[1, 1, 1] = fy; ez[1, 1, 0] = fz
    ox, oy, oz = og.edges2cellaverages(ex, ey, ez, vol)
    grad = ox + oy + oz
    want = {(0, 0, 0): x0*y0*z0*(fx+fz)/4, (1, 0, 0): x1*y0*z0*fz/4, (0, 1, 0): x0*y1*z0*(fx+fy+fz)/4,
            (1, 1, 0): x1*y1*z0*(fy+fz)/4, (0, 0, 1): x0*y0*z1*fx/4, (1, 0, 1): 0j, (0, 1, 1): x0*y1*z1*(fx+fy)/4,
            (1, 1, 1): x1*y1*z1*fy/4}
    for ijk, w in want.items():
        assert abs(grad[ijk] - w) <= 1e-14 * max(abs(w), 1), ijk


def test_misfit_and_gradient():
    from oracle import gradient as og
    g = load_golden("gradient.npz")
    h, vnC, vol, shp, off = _parts(g)
    mis, res = og.misfit(g['synthetic'], g['observed'], g['weights'])
    assert abs(mis / float(g['misfit']) - 1) < 1e-14
    grad = og.gradient_on_grid(vnC, vol, g['efield'], g['bfield'], g['smu0'])
    assert relerr(grad, g['grad']) < 1e-14
    # the residual source: strengths as Simulation._get_rfield forms them, spread by the (golden-pinned) source code
    import emg3d_amd as em
    grid = em.TensorMesh(h, origin=g['origin'])
    st = og.residual_strengths(res, g['weights'], g['smu0'])
    rf = 0
    for i in range(res.size):
        rf = rf + np.array(em.get_source_field(grid, g['rec'][:, i], float(g['freq']), strength=st[i]))
    assert relerr(rf, g['rfield']) < 1e-12
