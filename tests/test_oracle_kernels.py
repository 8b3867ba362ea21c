"""Pin the ORACLE (oracle/) against golden vectors captured from the reference
(tests/golden/make_golden.py) -- per-kernel level.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, relerr

TOL = 5e-12   # relative max-norm; the restatement only re-associates sums


@pytest.fixture(scope="module", params=["kernels_c128.npz", "kernels_f64.npz"])
def gold(request):
    return load_golden(request.param)


def _mesh(orc, g):
    return orc.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])


def _model(orc, g, p=''):
    return orc.VModel(g[p + 'eta_x'], g[p + 'eta_y'], g[p + 'eta_z'], g[p + 'zeta'])


def test_amat_x(oracle, gold):
    m = _mesh(oracle, gold)
    r = gold['s'].copy()
    oracle.amat_x(m.vnC, r, gold['e'], gold['eta_x'], gold['eta_y'], gold['eta_z'], gold['zeta'], *m.h)
    assert relerr(r, gold['amat_x_r']) < TOL
    md = _model(oracle, gold)
    assert abs(oracle.residual(m, md, gold['s'], gold['e'], True) / gold['residual_norm'] - 1) < 1e-13


@pytest.mark.parametrize("direction,name", [(0, 'gs'), (1, 'gs_x'), (2, 'gs_y'), (3, 'gs_z')])
@pytest.mark.parametrize("nu", [1, 2, 3])
def test_gauss_seidel(oracle, gold, direction, name, nu):
    m = _mesh(oracle, gold)
    e = gold['e'].copy()
    oracle.gauss_seidel(m.vnC, e, gold['s'], gold['eta_x'], gold['eta_y'], gold['eta_z'],
                        gold['zeta'], *m.h, nu, direction=direction)
    assert relerr(e, gold[f'{name}_nu{nu}']) < TOL


# The colour schedule (`order=1`: the ordering `bench.py` times) pinned to REFERENCE arithmetic: `kernels_colour.npz` holds the
# schedule replayed with the reference's own core.gauss_seidel* on 2 x 2 (x 2)-cell sub-grids (SURVEY App. E,
# tests/golden/make_golden.py::colour_fixture; the replay reproduces the reference's lexicographic sweep bit for bit).
_COLOUR_CASES = [('c128', 'kernels_c128.npz'), ('f64', 'kernels_f64.npz'), ('odd', None), ('long', None)]


def colour_case(col, tag, fname):
    if fname is None:
        return {k: col[f'{tag}_{k}'] for k in ('hx', 'hy', 'hz', 'e', 's', 'eta_x', 'eta_y', 'eta_z', 'zeta')}
    return load_golden(fname)


@pytest.mark.parametrize("tag,fname", _COLOUR_CASES)
@pytest.mark.parametrize("direction,name", [(0, 'gs'), (1, 'gs_x'), (2, 'gs_y'), (3, 'gs_z')])
@pytest.mark.parametrize("nu", [1, 2, 3])
def test_colour_sweep_vs_reference(oracle, tag, fname, direction, name, nu):
    col = load_golden('kernels_colour.npz')
    g = colour_case(col, tag, fname)
    e = g['e'].copy()
    oracle.gauss_seidel((g['hx'].size, g['hy'].size, g['hz'].size), e, g['s'], g['eta_x'], g['eta_y'], g['eta_z'],
                        g['zeta'], g['hx'], g['hy'], g['hz'], nu, direction=direction, order=1)
    assert relerr(e, col[f'{tag}_{name}_colour_nu{nu}']) < TOL


@pytest.mark.parametrize("lr_dir", range(8))
def test_smoothing_dispatch(oracle, gold, lr_dir):
    m = _mesh(oracle, gold)
    e = gold['e'].copy()
    oracle.smoothing(m, _model(oracle, gold), gold['s'], e, 2, lr_dir)
    assert relerr(e, gold[f'smoothing_lr{lr_dir}']) < TOL


@pytest.mark.parametrize("sc_dir", range(7))
def test_restriction(oracle, gold, sc_dir):
    m = _mesh(oracle, gold)
    md = _model(oracle, gold)
    cm, cmd, cs, ce = oracle.restriction(m, md, gold['s'], gold['res'], sc_dir)
    p = f'restrict{sc_dir}_'
    for a, c in enumerate('xyz'):
        np.testing.assert_allclose(cm.h[a], gold[p + 'ch' + c], rtol=1e-14)
    assert relerr(cmd.eta_x, gold[p + 'eta_x']) < 1e-14
    assert relerr(cmd.eta_y, gold[p + 'eta_y']) < 1e-14
    assert relerr(cmd.eta_z, gold[p + 'eta_z']) < 1e-14
    assert relerr(cmd.zeta, gold[p + 'zeta']) < 1e-14
    wx, wy, wz = oracle.get_restriction_weights(m, cm, sc_dir)
    for nm, w in (('wx', wx), ('wy', wy), ('wz', wz)):
        for q, a in zip('l0r', w):
            np.testing.assert_allclose(a, gold[p + nm + q], rtol=1e-13, atol=1e-15)
    assert relerr(cs, gold[p + 'csfield']) < TOL
    assert not ce.any()


@pytest.mark.parametrize("sc_dir", range(7))
def test_prolongation(oracle, gold, sc_dir):
    m = _mesh(oracle, gold)
    md = _model(oracle, gold)
    cm, _, _, _ = oracle.restriction(m, md, gold['s'], gold['res'], sc_dir)
    e = gold['e'].copy()
    oracle.prolongation(m, e, cm, gold[f'prolong{sc_dir}_ce'].copy(), sc_dir)
    assert relerr(e, gold[f'prolong{sc_dir}_e']) < TOL


# --- known-answer tests restated from the reference's own unit tests ---------
def test_solve_kat(oracle):
    """6x6 known answer of core.solve vs numpy.linalg.solve
    (reference tests/test_core.py:163-222 uses the same construction)."""
    rng = np.random.default_rng(3)
    for dtype in (np.float64, np.complex128):
        n = 6
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
        oracle.core_solve(amat.copy(), x)
        np.testing.assert_allclose(x, np.linalg.solve(full, b), rtol=1e-12)


def test_restrict_weights_kat(oracle):
    """Hand-computed numbers of reference tests/test_core.py:422-441."""
    edges = np.array([0., 500, 1200, 2000, 3000])
    width = edges[1:] - edges[:-1]
    centr = edges[:-1] + width / 2
    c_edges = edges[::2]
    c_width = c_edges[1:] - c_edges[:-1]
    c_centr = c_edges[:-1] + c_width / 2
    wl, w0, wr = oracle.restrict_weights(edges, centr, width, c_edges, c_centr, c_width)
    np.testing.assert_allclose(wl, [350 / 250, 250 / 600, 400 / 900])
    np.testing.assert_allclose(w0, [1., 1., 1.])
    np.testing.assert_allclose(wr, [350 / 600, 500 / 900, 400 / 500])


def test_blocks_to_amat_pattern(oracle):
    """Exact integer pattern of the band layout (reference
    tests/test_core.py:480-524 / docstring diagram core.py:1350-1370)."""
    n = 3
    amat = np.zeros(6 * (5 * n - 4)); bvec = np.zeros(5 * n - 4)
    middle = np.arange(1., 26); left = np.arange(101., 126); rhs = np.arange(1., 6)
    for im in range(n):
        oracle.blocks_to_amat(amat, bvec, middle, left, rhs * (im + 1), im, n)
    full = np.zeros((5 * n - 4, 5 * n - 4))
    for j in range(5 * n - 4):
        for i in range(j, min(5 * n - 4, j + 6)):
            full[i, j] = amat[i + 5 * j]
    M = middle.reshape(5, 5, order='F'); Lf = left.reshape(5, 5, order='F')
    assert np.array_equal(np.tril(full[:5, :5]), np.tril(M))
    assert np.array_equal(np.tril(full[5:10, 5:10]), np.tril(M))
    # left block of block-row 1: row 0 (cols 1..4) and the diagonal only are guaranteed
    assert np.array_equal(full[5, 1:5], Lf[0, 1:])
    assert np.array_equal(np.diag(full[5:10, 0:5])[1:], np.diag(Lf)[1:])
    # last point
    assert full[10, 10] == M[0, 0]
    assert np.array_equal(full[10, 6:10], Lf[0, 1:])
    assert np.array_equal(bvec, np.r_[rhs, 2 * rhs, 3])


def test_line_equals_point_on_two_cell_grids(oracle):
    """Line smoothers == point smoother when the transverse dims have 2 cells
    (property tested by reference tests/test_core.py:100-153)."""
    rng = np.random.default_rng(5)
    for direction, vnC in ((1, (8, 2, 2)), (2, (2, 8, 2)), (3, (2, 2, 8))):
        h = [rng.uniform(10, 30, n) for n in vnC]
        m = oracle.Mesh(h, (0, 0, 0))
        eta = [(-1j * rng.uniform(1, 2, vnC)).copy(order='F') for _ in range(3)]
        zeta = rng.uniform(1, 2, vnC).copy(order='F')
        s = (rng.standard_normal(m.nE) + 1j * rng.standard_normal(m.nE))
        oracle.ensure_pec(m, s)
        e1 = np.zeros(m.nE, dtype=complex); e2 = e1.copy()
        oracle.gauss_seidel(m.vnC, e1, s, *eta, zeta, *m.h, 1, direction=direction)
        # point smoother on a single-node-line grid needs many sweeps to solve the
        # line exactly; instead compare the line solve with a dense solve of the
        # same unknowns through the residual: after a line solve every equation
        # of the (single) line is satisfied.
        r = oracle.residual(m, oracle.VModel(*eta, zeta), s, e1)
        assert np.abs(r).max() < 1e-12 * np.abs(s).max()


def test_colour_order_is_order_independent(oracle):
    """4-colour (lines) / 8-colour (points) sweeps: lines of one colour are
    independent -> colour sweep == same sweep with the colour's
    lines visited in reversed order.  Checked through symmetry: running the
    coloured sweep on the mirrored problem gives the mirrored result only if no
    intra-colour dependency exists."""
    g = load_golden("kernels_c128.npz")
    m = oracle.Mesh([g['hx'], g['hy'], g['hz']], g['origin'])
    for direction in (0, 1, 2, 3):
        e = g['e'].copy()
        oracle.gauss_seidel(m.vnC, e, g['s'], g['eta_x'], g['eta_y'], g['eta_z'], g['zeta'],
                            *m.h, 2, direction=direction, order=1)
        assert np.isfinite(e).all()
        # coloured smoother is a different (but convergent) smoother: it must
        # reduce the residual like the lexicographic one does.
        md = oracle.VModel(g['eta_x'], g['eta_y'], g['eta_z'], g['zeta'])
        r0 = oracle.residual(m, md, g['s'], g['e'], True)
        r1 = oracle.residual(m, md, g['s'], e, True)
        assert r1 < r0


@pytest.mark.parametrize("dtype", [np.complex128, np.float64])
def test_colour_sweep_threads_are_bit_identical(oracle, dtype):
    """The oracle runs the lines / nodes of one colour on several host threads when asked to (``set_threads``; the tests do, to get
    through the 256^3 ... 448^3 checks in seconds): they are independent, so the result must be BIT-identical to one thread --
    all four smoothers, odd extents, more threads than rows."""
    rng = np.random.default_rng(3)
    shape = (65, 70, 67)          # (passes below 32 k blocks stay on one thread: par_for; these have 38 k ... 80 k)
    h = [rng.uniform(10., 50., n) for n in shape]
    nE = shape[0] * (shape[1] + 1) * (shape[2] + 1) + (shape[0] + 1) * shape[1] * (shape[2] + 1) + (shape[0] + 1) * (shape[1] + 1) * shape[2]
    cplx = dtype is np.complex128

    def rnd(n):
        return (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
    e0, s = rnd(nE), rnd(nE)
    eta = [np.asfortranarray((rng.uniform(1., 5., shape) * (1j if cplx else 1.)).astype(dtype)) for _ in range(3)]
    zeta = np.asfortranarray(rng.uniform(1., 2., shape))
    prev = oracle.set_threads(1)
    try:
        for direction in (0, 1, 2, 3):
            out = {}
            for nt in (1, 3, 16):
                oracle.set_threads(nt)
                e = e0.copy()
                oracle.gauss_seidel(shape, e, s, *eta, zeta, *h, 2, direction=direction, order=1)
                out[nt] = e
            assert np.array_equal(out[1], out[3]) and np.array_equal(out[1], out[16]), direction
            assert not np.array_equal(out[1], e0)
    finally:
        oracle.set_threads(prev)
