"""CPU: host-side logic of the drop-in surface -- fields, models, meshes, MGParameters, termination
and direction rules -- against the committed fixtures (inputs/outputs captured from the reference,
tests/golden/make_golden.py) and the known answers of the reference's own tests
(reference tests/test_solver.py:499-574, test_fields.py, test_models.py).  No GPU, no C-ABI calls."""
import itertools

import numpy as np
import pytest

from conftest import load_golden, relerr

from emg3d_amd import fields, meshes, models, shard, solver


def _grid(g, pre=''):
    return meshes.TensorMesh([g[pre + 'hx'], g[pre + 'hy'], g[pre + 'hz']], origin=g[pre + 'origin'])


# ------------------------------------------------------------------ fields / models / meshes
@pytest.mark.parametrize("name", ['point', 'point_lap', 'dipole', 'dipole_x'])
def test_source_field_matches_reference(name):
    """fields.get_source_field (reference emg3d/fields.py:446-730): point dipoles, finite dipoles,
    a dipole along one axis, Laplace domain."""
    g = load_golden("source_fields.npz")
    grid = _grid(g)
    s = fields.get_source_field(grid, g[f'{name}_src'], float(g[f'{name}_freq']))
    ref = g[f'{name}_sfield']
    assert s.dtype == ref.dtype
    assert relerr(s, ref) < 1e-15
    assert abs(s.smu0 - g[f'{name}_smu0']) <= 1e-16 * abs(g[f'{name}_smu0'])
    # the real, frequency-independent source vector (fields.py: SourceField.vector)
    assert s.vector.dtype == np.float64
    assert relerr(s.smu0 * s.vector, np.asarray(s)) < 1e-15


def _source_kwargs(g, name):
    st = g[f'{name}_strength']
    return dict(strength=complex(st) if np.iscomplexobj(st) else float(st), electric=bool(g[f'{name}_electric']),
                length=float(g[f'{name}_length']))


@pytest.mark.parametrize("name", ['shaped', 'shaped_strength', 'loop', 'loop_big', 'shaped_mag', 'point_len'] +
                         [f'oblique_{i}' for i in range(6)])
def test_shaped_and_magnetic_sources_match_reference(name):
    """Arbitrarily shaped sources (summed segments, fields.py:555-577), magnetic point dipoles (square loop of electric
    dipoles, negated: fields.py:547-549, 574-576, 1043-1049), strength and dipole length."""
    g = load_golden("source_fields.npz")
    grid = _grid(g)
    s = fields.get_source_field(grid, g[f'{name}_src'], float(g[f'{name}_freq']), **_source_kwargs(g, name))
    ref = g[f'{name}_sfield']
    assert s.dtype == ref.dtype
    assert relerr(s, ref) < 1e-15
    np.testing.assert_allclose(s.moment, g[f'{name}_moment'], rtol=1e-14, atol=1e-14)


def test_source_field_errors():
    g = load_golden("source_fields.npz")
    grid = _grid(g)
    with pytest.raises(ValueError, match="Source is wrong defined"):
        fields.get_source_field(grid, [0., 0., 0., 0.], 1.0)
    with pytest.raises(ValueError, match="same dimension"):
        fields.get_source_field(grid, [[0., 1.], [0., 1., 2.], [0., 1.]], 1.0)
    with pytest.raises(ValueError, match="no length"):
        fields.get_source_field(grid, [1., 1., 2., 2., 3., 3.], 1.0)
    with pytest.raises(ValueError, match="requires the frequency"):
        fields.SourceField(grid)


def test_field_views_and_pec():
    """Field = ONE 1-D buffer [fx, fy, fz], F-ordered views (reference fields.py:253-281, 341-360)."""
    grid = meshes.TensorMesh([[1., 2., 3.], [1., 1.], [2., 2., 2., 2.]], origin=(0, 0, 0))
    assert (grid.nEx, grid.nEy, grid.nEz) == (3 * 3 * 5, 4 * 2 * 5, 4 * 3 * 4)
    f = fields.Field(grid, np.arange(grid.nE, dtype=float) + 1.0, freq=-1.0)
    assert f.fx.shape == (3, 3, 5) and f.fy.shape == (4, 2, 5) and f.fz.shape == (4, 3, 4)
    assert f.fx.flags.f_contiguous and np.shares_memory(f.fx, f.field)
    assert f.fy.ravel('F')[0] == grid.nEx + 1 and f.fz.ravel('F')[0] == grid.nEx + grid.nEy + 1
    f.ensure_pec
    assert not f.fx[:, [0, -1], :].any() and not f.fx[:, :, [0, -1]].any() and f.fx[:, 1, 1:-1].all()
    assert not f.fy[[0, -1], :, :].any() and not f.fz[:, [0, -1], :].any()
    c = f.copy()
    c.field[:] = 0
    assert f.field.any()
    # Field keeps the requested dtype (default complex); SourceField follows the domain (fields.py:417-420)
    assert fields.Field(grid, freq=-2.0).dtype == np.complex128 and fields.Field(grid, dtype=float).dtype == np.float64
    assert fields.SourceField(grid, freq=2.0).dtype == np.complex128 and fields.SourceField(grid, freq=-2.0).dtype == np.float64
    with pytest.raises(ValueError, match='`freq` must be >0'):
        fields.Field(grid, freq=0.0)


def test_volume_model_matches_reference():
    """models.VolumeModel (reference emg3d/models.py:554-658): eta = s mu_0 sigma V, zeta = V / mu_r,
    aliasing of eta_y/eta_z; and the frequency-independent split models.sigma_volume."""
    g = load_golden("solves_16.npz")
    grid = _grid(g)
    model = models.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = fields.get_source_field(grid, g['src'], float(g['freq']))
    assert relerr(sfield, g['sfield']) < 1e-15
    vm = models.VolumeModel(grid, model, sfield)
    assert relerr(vm.eta_x, g['eta_x']) < 1e-15 and relerr(vm.zeta, g['zeta']) < 1e-15
    assert relerr(vm.eta_y, g['eta_x'] / 2) < 1e-15 and relerr(vm.eta_z, g['eta_x'] / 3) < 1e-15
    sv = models.sigma_volume(grid, model)
    for sv_c, eta_c in zip(sv[:3], (vm.eta_x, vm.eta_y, vm.eta_z)):
        assert sv_c.dtype == np.float64 and relerr(sfield.smu0 * sv_c, eta_c) < 1e-15
    assert np.array_equal(sv[3], vm.zeta)
    iso = models.Model(grid, g['rho_b'])
    vmi = models.VolumeModel(grid, iso, sfield)
    assert vmi.eta_y is vmi.eta_x and vmi.eta_z is vmi.eta_x
    svi = models.sigma_volume(grid, iso)
    assert svi[1] is svi[0] and svi[2] is svi[0]
    with pytest.raises(ValueError, match="epsilon_r"):
        models.sigma_volume(grid, models.Model(grid, g['rho_b'], epsilon_r=np.ones(grid.nC)))
    g2 = load_golden("regression.npz")
    grid2 = _grid(g2, 'res_')
    m2 = models.Model(grid2, g2['res_property_x'], g2['res_property_y'], g2['res_property_z'])
    s2 = fields.SourceField(grid2, g2['res_sfield'].copy(), freq=float(g2['res_freq']))
    assert abs(s2.smu0 - g2['res_smu0_here']) <= 1e-16 * abs(g2['res_smu0_here'])
    assert relerr(models.VolumeModel(grid2, m2, s2).eta_x, g2['res_eta_x_here']) < 1e-15


def test_eta_factored_is_bit_identical():
    """models.eta_factored: alpha * (real array) equals VolumeModel's eta bit for bit (frequency and
    Laplace domain, tri-axial and isotropic), None with epsilon_r (that factorisation does not exist then; solve() uses
    models.model_parts) -- the sigma*V handles are built from it."""
    g = load_golden("regression.npz")
    for pre in ('res_', 'lap_'):
        grid = _grid(g, pre)
        m = models.Model(grid, g[pre + 'property_x'], g[pre + 'property_y'], g[pre + 'property_z'])
        s = fields.SourceField(grid, g[pre + 'sfield'].copy(), freq=float(g[pre + 'freq']))
        vm = models.VolumeModel(grid, m, s)
        sx, sy, sz, zeta, alpha = models.eta_factored(grid, m, s)
        assert alpha == (1j if pre == 'res_' else 1.0)
        for sv, eta in zip((sx, sy, sz), (vm.eta_x, vm.eta_y, vm.eta_z)):
            assert sv.dtype == np.float64 and np.array_equal(alpha * sv, eta)
        assert np.array_equal(zeta, vm.zeta)
    g = load_golden("solves_16.npz")
    grid = _grid(g)
    s = fields.get_source_field(grid, g['src'], float(g['freq']))
    iso = models.eta_factored(grid, models.Model(grid, g['rho_b'], mu_r=1.5 * np.ones(grid.nC)), s)
    assert iso[1] is iso[0] and iso[2] is iso[0]
    assert np.array_equal(iso[3], models.VolumeModel(grid, models.Model(grid, g['rho_b'], mu_r=1.5 * np.ones(grid.nC)), s).zeta)
    vti = models.eta_factored(grid, models.Model(grid, g['rho_b'], property_z=2 * g['rho_b']), s)
    assert vti[1] is vti[0] and vti[2] is not vti[0]
    assert models.eta_factored(grid, models.Model(grid, g['rho_b'], epsilon_r=np.ones(grid.nC)), s) is None


def test_stretched_widths_and_mesh():
    h = meshes.stretched_widths(4, 2, 50., 1.2)        # reference tests/test_meshes.py:28-31
    np.testing.assert_allclose(h, [72., 60., 50., 50., 50., 50., 60., 72.])
    grid = meshes.TensorMesh([h, h[:4], h[:2]], origin=(-10., 0., 5.))
    assert tuple(grid.vnC) == (8, 4, 2) and grid.nC == 64
    np.testing.assert_allclose(grid.nodes_x[:3], [-10., 62., 122.])
    np.testing.assert_allclose(grid.cell_centers_z, [5 + 36., 5 + 72 + 30.])
    np.testing.assert_allclose(grid.cell_volumes.reshape(grid.vnC, order='F')[1, 2, 0], 60. * 50. * 72.)


# ------------------------------------------------------------------ MGParameters (reference tests/test_solver.py:499-574)
def test_mgparameters_known_answers():
    vnC = (2**3, 2**5, 2**4)
    mk = lambda **kw: solver.MGParameters(**dict(dict(cycle='F', sslsolver=False, semicoarsening=False,
                                                      linerelaxation=False, vnC=vnC, verb=1), **kw))
    assert 'semicoarsening : True [1 2 3]' in repr(mk(semicoarsening=True))
    assert 'semicoarsening : True [1 2 1 3]' in repr(mk(cycle='V', semicoarsening=1213))
    assert 'semicoarsening : True [2]' in repr(mk(semicoarsening=2))
    with pytest.raises(ValueError, match='`semicoarsening` must be one of'):
        mk(semicoarsening=5)
    assert 'linerelaxation : True [4 5 6]' in repr(mk(linerelaxation=True))
    assert 'linerelaxation : True [1 2 4 7]' in repr(mk(linerelaxation=1247))
    var = mk(linerelaxation=1, clevel=1)
    assert 'linerelaxation : True [1]' in repr(var)
    np.testing.assert_allclose(var.clevel, 1)
    with pytest.raises(ValueError, match='`linerelaxation` must be one of'):
        mk(linerelaxation=-9)
    with pytest.raises(ValueError, match='At least `cycle` or `sslsolver`'):
        mk(cycle=None)
    var = mk(sslsolver=True, semicoarsening=True, maxit=33)
    assert "sslsolver : 'bicgstab'" in repr(var)
    assert var.ssl_maxit == 33 and var.maxit == 3
    for bad in ('abcd', 4):
        with pytest.raises(ValueError, match='`sslsolver` must be True'):
            mk(sslsolver=bad)
    with pytest.raises(ValueError, match='`cycle` must be one of'):
        mk(cycle='G')
    with pytest.raises(ValueError, match='Nr. of cells must be at least'):
        mk(vnC=(1, 2, 3))
    txt = ":: Grid not optimal for MG solver ::"
    assert txt in repr(mk(vnC=(11 * 2**3, 2**5, 2**4)))
    assert txt not in repr(mk(vnC=(11 * 2**5, 11 * 2**4, 11 * 2**5), clevel=4))
    assert txt in repr(mk(vnC=(11 * 2**5, 11 * 2**4, 11 * 2**5), clevel=5))
    assert txt in repr(mk(vnC=(2**3, 2**3, 2**3)))
    with pytest.raises(ValueError, match='`ordering` must be one of'):
        mk(ordering='redblack')


def test_mgparameters_levels_and_rotation():
    """clevel per semicoarsening direction (solver.py:1142-1206) and the sc/lr rotation iterators."""
    var = solver.MGParameters(cycle='F', sslsolver=False, semicoarsening=True, linerelaxation=True,
                              vnC=(128, 128, 128), verb=0)
    assert list(var.clevel) == [6, 6, 6, 6]
    assert [var.sc_dir] + [next(var.sc_cycle) for _ in range(5)] == [1, 2, 3, 1, 2, 3]
    assert [var.lr_dir] + [next(var.lr_cycle) for _ in range(5)] == [4, 5, 6, 4, 5, 6]
    var = solver.MGParameters(cycle='V', sslsolver=False, semicoarsening=False, linerelaxation=False,
                              vnC=(48, 24, 20), verb=0)
    # 48 = 3*2^4, 24 = 3*2^3, 20 = 5*2^2: coarsening stops at the odd factors
    assert list(var.clevel) == [4, 3, 4, 4] and var.sc_dir == 0 and var.lr_dir == 0 and var.cycmax == 1
    assert solver.MGParameters(cycle='W', sslsolver=False, semicoarsening=False, linerelaxation=False,
                               vnC=(8, 8, 8), verb=0).cycmax == 2


class _G:
    def __init__(self, vnC):
        self.vnC = np.array(vnC)


def test_current_directions():
    """_current_sc_dir / _current_lr_dir (reference solver.py:1467-1572): every requested direction on
    grids with odd / 2-cell dimensions; must equal the device-side rules (csrc/mg.hpp)."""
    assert [solver._current_sc_dir(d, _G((8, 8, 8))) for d in range(4)] == [0, 1, 2, 3]
    assert solver._current_sc_dir(0, _G((3, 8, 8))) == 1          # x cannot be coarsened
    assert solver._current_sc_dir(2, _G((3, 8, 8))) == 6          # x and y kept
    assert solver._current_sc_dir(3, _G((3, 8, 8))) == 5          # x and z kept
    assert solver._current_sc_dir(0, _G((2, 2, 8))) == 6
    assert solver._current_sc_dir(1, _G((8, 5, 8))) == 6 and solver._current_sc_dir(3, _G((8, 5, 8))) == 4
    for lr in range(8):
        assert solver._current_lr_dir(lr, _G((8, 8, 8))) == lr
    assert [solver._current_lr_dir(lr, _G((2, 8, 8))) for lr in range(8)] == [0, 0, 2, 3, 4, 3, 2, 4]
    assert [solver._current_lr_dir(lr, _G((8, 2, 8))) for lr in range(8)] == [0, 1, 0, 3, 3, 5, 1, 5]
    assert [solver._current_lr_dir(lr, _G((8, 8, 2))) for lr in range(8)] == [0, 1, 2, 0, 2, 1, 6, 6]
    assert solver._current_lr_dir(7, _G((2, 2, 8))) == 3


def test_terminate_order():
    """_terminate (reference solver.py:1682-1744): converged / diverged / stagnated / maxit, in this
    order, and the abort of the Krylov driver."""
    var = solver.MGParameters(cycle='F', sslsolver=False, semicoarsening=False, linerelaxation=False,
                              vnC=(8, 8, 8), verb=0, maxit=5, tol=1e-3)
    var.l2_refe = 1.0
    assert solver._terminate(var, 1e-4, 1.0, 1) and var.exit_message == "CONVERGED"
    assert solver._terminate(var, 11.0, 1.0, 1) and var.exit_message == "DIVERGED"
    assert solver._terminate(var, np.nan, 1.0, 1) and var.exit_message == "DIVERGED"
    assert not solver._terminate(var, 0.5, 0.4, 2)                       # stagnation only after 2 cycles
    assert solver._terminate(var, 0.5, 0.4, 3) and var.exit_message == "STAGNATED"
    assert solver._terminate(var, 0.5, 0.6, 5) and var.exit_message.startswith("MAX. ITERATION")
    assert not solver._terminate(var, 0.5, 0.6, 4)
    kv = solver.MGParameters(cycle='F', sslsolver=True, semicoarsening=False, linerelaxation=False,
                             vnC=(8, 8, 8), verb=0)
    kv.l2_refe = 1.0
    with pytest.raises(solver._ConvergenceError):
        solver._terminate(kv, 11.0, 1.0, 1)
    assert solver._terminate(kv, 0.5, 0.6, kv.maxit)                     # maxit of the preconditioner: no abort


def test_my_frequencies():
    f = [0.25, 0.5, 0.75, 1, 1.5, 2, 3, 4]
    assert shard.my_frequencies(f, 0, 8) == [0.25] and shard.my_frequencies(f, 7, 8) == [4.0]
    assert shard.my_frequencies(f, 1, 3) == [0.5, 1.5, 4.0]
    assert sorted(itertools.chain(*[shard.my_frequencies(f, r, 3) for r in range(3)])) == sorted(map(float, f))


def test_regular_grid_prolongator():
    """RegularGridProlongator / _get_prolongation_coordinates (reference solver.py:1368-1463, 1841-1845;
    reference test tests/test_solver.py:577-640 compares with SciPy's RegularGridInterpolator): bilinear
    inside, linear extrapolation outside, weights computed once."""
    import scipy.interpolate as si
    rng = np.random.default_rng(2)
    cgrid = meshes.TensorMesh([[1.], rng.uniform(1, 3, 6), rng.uniform(1, 3, 4)], origin=(0., -2., 1.))
    fgrid = meshes.TensorMesh([[1.], np.repeat(cgrid.h[1], 2) / 2, np.repeat(cgrid.h[2], 2) / 2], origin=(0., -2., 1.))
    pts = solver._get_prolongation_coordinates(fgrid, 'y', 'z')
    assert pts.shape == ((fgrid.vnC[1] + 1) * (fgrid.vnC[2] + 1), 2)
    np.testing.assert_allclose(pts[:3, 0], fgrid.nodes_y[:3])
    np.testing.assert_allclose(pts[:3, 1], fgrid.nodes_z[0])
    np.testing.assert_allclose(pts[fgrid.vnC[1] + 1], [fgrid.nodes_y[0], fgrid.nodes_z[1]])
    vals = rng.standard_normal((cgrid.vnC[1] + 1, cgrid.vnC[2] + 1)) + 1j * rng.standard_normal((cgrid.vnC[1] + 1, cgrid.vnC[2] + 1))
    fn = solver.RegularGridProlongator(cgrid.nodes_y, cgrid.nodes_z, pts)
    assert fn.size == pts.shape[0] and fn.weight.shape == (4, pts.shape[0])
    ref = si.RegularGridInterpolator((cgrid.nodes_y, cgrid.nodes_z), vals, bounds_error=False, fill_value=None)(pts)
    np.testing.assert_allclose(fn(vals), ref, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(fn.weight.sum(axis=0), 1.0, rtol=1e-14)
    # points outside the coarse grid: linear extrapolation from the last interval, like SciPy's fill_value=None
    out = np.array([[cgrid.nodes_y[0] - 1.0, cgrid.nodes_z[-1] + 0.5], [cgrid.nodes_y[-1] + 2.0, cgrid.nodes_z[0] - 0.25]])
    f2 = solver.RegularGridProlongator(cgrid.nodes_y, cgrid.nodes_z, out)
    ref2 = si.RegularGridInterpolator((cgrid.nodes_y, cgrid.nodes_z), vals.real, bounds_error=False, fill_value=None)(out)
    np.testing.assert_allclose(f2(vals.real), ref2, rtol=1e-12)
    # a linear function is reproduced exactly, also when called repeatedly (weights are reused)
    Y, Z = np.meshgrid(cgrid.nodes_y, cgrid.nodes_z, indexing='ij')
    lin = 2.0 * Y - 3.0 * Z + 1.0
    for _ in range(2):
        np.testing.assert_allclose(fn(lin), 2.0 * pts[:, 0] - 3.0 * pts[:, 1] + 1.0, rtol=1e-13, atol=1e-13)


def test_bench_spawns_ranks():
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks
    itself as fresh child processes with the torch.distributed rank environment, relays rank 0's JSON line
    and exits non-zero when a rank fails.  `--echo-env` is the harness self-test: no GPU is touched."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMP_NUM_THREADS",
                                                            "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--echo-env"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["rank"] == 0 and line["world"] == 3 and line["master"] == "127.0.0.1" and int(line["port"]) > 0
    # every rank's host BLAS / OpenMP pools are pinned to one thread before the rank imports NumPy
    assert all(v == "1" for v in line["threads"].values()) and "OPENBLAS_NUM_THREADS" in line["threads"]
    p = subprocess.run(cmd + ["--fail-rank", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "ranks failed" in p.stderr
    # under a launcher (WORLD_SIZE set) nothing is spawned: the process IS the rank
    p = subprocess.run(cmd, env=dict(env, WORLD_SIZE="3", RANK="1", LOCAL_RANK="1"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == ""


def test_frequency_spec_matches_source_field():
    """fields.FrequencySpec stands in for a SourceField where only the frequency matters (source built in HBM)."""
    import emg3d_amd as em
    from emg3d_amd.fields import FrequencySpec
    grid = em.TensorMesh([np.ones(4), np.ones(3), np.ones(2)], origin=(0., 0., 0.))
    for f in (1.5, -2.0):
        sf, fs = em.SourceField(grid, freq=f), FrequencySpec(f)
        assert fs.dtype == sf.dtype and fs._freq == sf._freq and fs.freq == sf.freq
        assert fs.smu0 == sf.smu0 and fs.sval == sf.sval
    with pytest.raises(ValueError, match="must be >0"):
        FrequencySpec(0.0)


def test_field_like_the_reference(tmp_path):
    """The checks of the reference's tests/test_fields.py:test_field / test_source_field on the container types."""
    import shelve
    from scipy import constants
    grid = meshes.TensorMesh([np.array([.5, 8]), np.array([1, 4]), np.array([2, 8])], np.zeros(3))
    rng = np.random.default_rng(0)

    def dummy(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    ex, ey, ez = dummy(*grid.vnEx), dummy(*grid.vnEy), dummy(*grid.vnEz)
    ee = fields.Field(ex, ey, ez)
    np.testing.assert_allclose(ee, np.r_[ex.ravel('F'), ey.ravel('F'), ez.ravel('F')])
    np.testing.assert_allclose(ee.fx, ex); np.testing.assert_allclose(ee.fy, ey); np.testing.assert_allclose(ee.fz, ez)
    assert ee.smu0 is None and ee.sval is None and ee.is_electric is True
    np.testing.assert_allclose(ee.fx.amp(), np.abs(ee.fx))
    np.testing.assert_allclose(ee.fy.pha(unwrap=False), np.angle(ee.fy))
    ee2 = fields.Field(grid, ee.field)
    np.testing.assert_allclose(ee.field, ee2.field); np.testing.assert_allclose(ee.fx, ee2.fx)
    ee3 = fields.Field(grid)
    assert ee.shape == ee3.shape
    ee3.field = ee.field
    np.testing.assert_allclose(ee.field, ee3.field)
    ee3.fx = ee.fx; ee3.fy = ee.fy; ee3.fz = ee.fz
    np.testing.assert_allclose(ee.field, ee3.field)
    ee.ensure_pec
    assert abs(np.sum(ee.fx[:, 0, :] + ee.fx[:, -1, :])) == 0 and abs(np.sum(ee.fz[:, 0, :] + ee.fz[:, -1, :])) == 0
    e2 = ee.copy()
    np.testing.assert_allclose(ee.field, e2.field)
    assert ee.field.base is not e2.field.base
    edict = ee.to_dict()
    np.testing.assert_allclose(fields.Field.from_dict(edict), ee)
    del edict['field']
    with pytest.raises(KeyError, match="Variable 'field' missing"):
        fields.Field.from_dict(edict)
    grid.nEx = None
    with pytest.raises(ValueError, match='Provided grid must be a 3D grid'):
        fields.Field(grid)
    with shelve.open(str(tmp_path / 'test')) as db:
        db['field'] = ee2
    with shelve.open(str(tmp_path / 'test')) as db:
        test = db['field']
    np.testing.assert_allclose(test, ee2)
    # SourceField
    grid = meshes.TensorMesh([np.array([.5, 8]), np.array([1, 4]), np.array([2, 8])], np.zeros(3))
    ss = fields.SourceField(grid, freq=np.pi)
    np.testing.assert_allclose(ss.smu0, -2j * np.pi * np.pi * constants.mu_0)
    assert hasattr(ss, 'vector') and hasattr(ss, 'vx') and ss.vy.shape == grid.vnEy
    with pytest.raises(ValueError, match='`freq` must be >0'):
        fields.SourceField(grid, freq=0)


def test_source_field_like_the_reference():
    """reference tests/test_fields.py: test_get_source_field, test_arbitrarily_shaped_source,
    test_get_source_field_point_vs_finite on the host twin of get_source_field."""
    import _source_checks
    _source_checks.run(fields.get_source_field, meshes, fields)


def test_model_parts_reproduce_volume_model_eta():
    """models.model_parts: (sigma, V, zeta) from which the device forms eta (emg3d_mg_create_vs).  The same two products
    in NumPy -- (b V) sigma, times i in the frequency domain -- give VolumeModel's eta bit for bit (reference
    models.py:631-658: `(smu0 * vol) * sigma`), for tri-axial, VTI and isotropic models, with mu_r, both mappings."""
    import emg3d_amd as em
    from emg3d_amd import models
    rng = np.random.default_rng(5)
    grid = em.TensorMesh([rng.uniform(10, 50, 6), rng.uniform(10, 50, 5), rng.uniform(10, 50, 4)], origin=(0, 0, 0))
    rho = 10 ** rng.uniform(-1, 2, grid.vnC)
    for kw in (dict(property_x=rho), dict(property_x=rho, property_z=3 * rho), dict(property_x=rho, property_y=2 * rho, property_z=3 * rho),
               dict(property_x=rho, mu_r=1 + rng.uniform(0, 2, grid.vnC)), dict(property_x=1 / rho, mapping='Conductivity')):
        model = em.Model(grid, **kw)
        sx, sy, sz, vol, zeta = models.model_parts(grid, model)
        assert (sy is sx) == (model.case in (0, 2)) and (sz is sx) == (model.case in (0, 1))
        for freq in (1.3, -2.0):
            sf = em.SourceField(grid, freq=freq)
            vm = em.VolumeModel(grid, model, sf)
            b = np.imag(sf.smu0) if np.iscomplexobj(sf.smu0) else float(sf.smu0)
            for sig, eta in ((sx, vm.eta_x), (sy, vm.eta_y), (sz, vm.eta_z)):
                t = (b * vol) * sig
                want = np.asarray(eta)
                if np.iscomplexobj(want):
                    assert np.array_equal(want.imag, t) and not want.real.any()
                else:
                    assert np.array_equal(want, t)
            assert np.array_equal(np.asarray(vm.zeta), zeta)
    # with epsilon_r: the same parts plus the permittivities; eta = (b V)(c eps_r) + i (b V) sigma resp. (b V)(sigma - c eps_r)
    # with c = models.seps0_of(s) is what VolumeModel (NumPy's complex array expressions) gives, bit for bit -- the
    # arithmetic k_eta_vs_eps runs on the device
    eps = 10 ** rng.uniform(4.5, 6.5, grid.vnC)
    model = em.Model(grid, rho, 2 * rho, 3 * rho, epsilon_r=eps)
    parts = models.model_parts(grid, model)
    sx, sy, sz, vol, zeta = parts
    assert np.array_equal(parts.epsilon_r, eps) and models.model_parts(grid, em.Model(grid, rho)).epsilon_r is None
    for freq in (1.3, 20.0, -2.0):
        sf = em.SourceField(grid, freq=freq)
        vm = em.VolumeModel(grid, model, sf)
        b = np.imag(sf.smu0) if np.iscomplexobj(sf.smu0) else float(sf.smu0)
        c = models.seps0_of(sf.sval)
        for sig, eta in ((sx, vm.eta_x), (sy, vm.eta_y), (sz, vm.eta_z)):
            want = np.asarray(eta)
            p, t = b * vol, c * parts.epsilon_r
            if np.iscomplexobj(want):
                assert np.array_equal(want.real, p * t) and np.array_equal(want.imag, p * sig) and want.real.any()
            else:
                assert np.array_equal(want, p * (sig - t))


def test_bench_cycle_algorithmic_bytes():
    """bench.cycle_alg_bytes: the whole-cycle algorithmic byte count behind `cycle_algorithmic.frac`.  Hand count for an
    8 x 8 x 8 V-cycle without rotation effects (all three (sc, lr) states are equivalent on a cube): levels 8^3 ->
    (sc_dir 1: x kept) 8x4x4 -> 8x2x2 (coarsest for sc_dir 1: clevel = 2)."""
    import bench
    got = bench.cycle_alg_bytes((8, 8, 8), 'V')
    c0, c1, c2 = 512, 128, 32
    # level 0, 1: (2 + 2) sweeps x 2 line directions x 200 + residual 200 + restriction 54 + prolongation 102 per cell
    per = 4 * 2 * 200 + 200 + 54 + 102
    # coarsest 8x2x2 with lr_dir 4 (y, z lines) degrades to the point smoother on the two 2-cell axes: 1 sweep x 1
    want = per * c0 + per * c1 + 1 * 1 * 200 * c2 + 200 * c0
    assert got == want, (got, want)
    # F-cycle: level l is visited more often than in a V-cycle, never less
    assert bench.cycle_alg_bytes((32, 32, 32), 'F') > bench.cycle_alg_bytes((32, 32, 32), 'V')


def test_model_gradient_sign_and_chain():
    """optimize.model_gradient = the reference's last two steps of optimize.gradient (optimize.py:201-214, gridding
    'same'): d(misfit)/d(sigma) = -grad; resistivity models get MapResistivity.derivative_chain's factor -1/rho^2."""
    import emg3d_amd as em
    from emg3d_amd import optimize
    grid = em.TensorMesh([np.ones(3), np.ones(2), np.ones(2)], origin=(0., 0., 0.))
    rng = np.random.default_rng(4)
    rho = rng.uniform(1, 10, grid.nC)
    g = rng.standard_normal(grid.vnC)
    mc = em.Model(grid, 1 / rho, mapping='Conductivity')
    mr = em.Model(grid, rho, mapping='Resistivity')
    assert np.array_equal(optimize.model_gradient(grid, mc, g), -g)
    want = -g * (-(1.0 / rho.reshape(grid.vnC, order='F')) ** 2)
    np.testing.assert_allclose(optimize.model_gradient(grid, mr, g), want, rtol=1e-15)


def _figure_of(log):
    """The cycle-QC block of a verb=4 log: from the '       h_' line up to (not including) the first iteration line."""
    lines = str(log).split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith("       h_"))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith("   [") and "after" in lines[i])
    return "\n".join(lines[i0:i1]) + "\n"


def test_cycle_qc_figure_equals_reference():
    """The ASCII cycle picture of the reference's verb > 3 log (emg3d/solver.py:1603-1632), rebuilt from the V/W/F rule on
    level numbers (the recursion itself runs on the device): identical text for V / W / F cycles with and without
    semicoarsening, a capped `clevel`, a 64 x 4 x 4 W-cycle (> 70 steps: truncated with the reference's note) and a
    2 x 2 x 2 grid (no coarse level).  Fixture: tests/golden/logs.npz, generated by running the reference."""
    import ast
    from emg3d_amd.solver import MGParameters, _cycle_qc_figure, _first_cycle_levels
    g = load_golden("logs.npz")
    for tag in g['cases']:
        kw = ast.literal_eval(str(g[f'{tag}_kw']))
        var = MGParameters(verb=4, vnC=tuple(int(n) for n in g[f'{tag}_shape']), sslsolver=False,
                           semicoarsening=kw.get('semicoarsening', False), linerelaxation=kw.get('linerelaxation', False),
                           cycle=kw['cycle'], clevel=kw.get('clevel', -1))
        want = _figure_of(g[f'{tag}_log'])
        got = _cycle_qc_figure(_first_cycle_levels(var))
        assert got == want, (str(tag), got, want)


def test_wrap_coordinates_follow_scipy():
    """maps._wrap_coords restates scipy.ndimage's coordinate rule of mode 'wrap' (period n - 1); with the mirror spline
    it reproduces map_coordinates(mode='wrap') -- checked here in one dimension on the host (the device evaluation is
    checked against the reference's outputs in tests/test_gpu_receivers.py)."""
    from scipy import ndimage
    from emg3d_amd import maps
    rng = np.random.default_rng(3)
    for n in (4, 9, 10):
        a = rng.standard_normal(n)
        coef = ndimage.spline_filter1d(a, order=3, mode='mirror')
        x = np.concatenate([rng.uniform(-300, 300, 500), [0., n - 1., -1., float(n), -(n - 1.), 2 * (n - 1.)]])
        ref = ndimage.map_coordinates(a, [x], order=3, mode='wrap')
        c = maps._wrap_coords(x, n)
        assert (c >= 0).all() and (c <= n - 1).all()
        fl = np.floor(c)
        t = c - fl
        w = np.stack([(1 - t) ** 3 / 6, (3 * t ** 3 - 6 * t ** 2 + 4) / 6, (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6, t ** 3 / 6])
        idx = fl.astype(int)[None, :] - 1 + np.arange(4)[:, None]
        s2 = 2 * n - 2
        idx = np.abs(idx) % s2                         # mirror: period 2 n - 2, symmetric about 0
        idx = np.where(idx >= n, s2 - idx, idx)
        got = (w * coef[idx]).sum(axis=0)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_colour_schedules_of_device_and_oracle_agree():
    """The visiting order of the line / point colours is this build's own choice (profiles/HISTORY.md A.12); the device path
    (MG::colour_perm, colour_perm_b, point_perm) and the oracle's colour twin (gs_line, gs_point) must carry the same
    tables -- the GPU parity tests compare the two, this guards the sources against drifting apart unnoticed on a CPU box."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mg = open(os.path.join(root, "emg3d_amd", "csrc", "mg.hpp")).read()
    orc = open(os.path.join(root, "oracle", "emg3d_oracle.cpp")).read()

    def table(text, name):
        m = re.search(name + r"\[\d\]\s*=\s*\{([\d,\s]+)\}", text)
        assert m, name
        return [int(v) for v in m.group(1).split(",")]

    assert table(mg, "colour_perm") == table(orc, "kColourFwd") == [1, 3, 0, 2]
    assert table(mg, "colour_perm_b") == table(orc, "kColourBwd") == [0, 3, 2, 1]
    assert table(mg, "point_perm") == table(mg, "point_perm_b") == list(range(8))
    assert "const int col = ch;" in orc          # gs_point: 0..7 in every sweep


def test_bench_cpu_baseline_keys():
    """bench.cpu_baseline: single-thread leg + the `concurrent` leg (BASELINE configs[4] on the host: min(8, cores) single-thread
    solves side by side), CPU model string, the -march it was compiled for.  Small workload; host code only."""
    import bench
    import emg3d_amd as em
    cb = bench.cpu_baseline(em, "32F")
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"].startswith("Mcells/s")
    assert isinstance(cb["cpu_model"], str) and cb["cpu_model"]
    assert cb["march"].startswith("native") or cb["march"].startswith("x86-64-v3")
    nw = min(8, bench._cpu_cores())
    if nw > 1:
        cc = cb["concurrent"]
        assert cc["workers"] == nw == cc["cores"] == len(cc["per_worker"]) and cc["freqs_Hz"] == bench.FREQS[:nw]
        assert abs(cc["value"] - sum(cc["per_worker"])) < 1e-9 * cc["value"] and "862-867" in cc["sample"]


def test_bench_roofline_sampling():
    """bench.roofline_of: SWEEP_SAMPLES event brackets per direction, `launch_ms` = the median of the per-sample launch means,
    min / median / max beside it; dense and dipole source."""
    import bench

    class Dev:
        def __init__(self):
            self.k = 0
            self.dense = False

        def time_sweep(self, d, reps):
            self.k += 1
            return (8.0 if self.dense else 4.0) * (1.0 + 0.01 * (self.k % 5))      # ms per sweep of 4 launches

        def set_sfield(self, s):
            self.dense = not self.dense

        def last_sweep_kernel(self):
            return "k_line_sweep_qc<c128,3,16>"

        def placement(self):
            return {"x": {"tries": 3, "kept": 2, "first_ms": 2.6, "kept_ms": 2.3, "ms_per_sweep": [2.6, 2.5, 2.3]},
                    "yz": {"tries": 0, "kept": -1, "reused": True}}

    class Grid:
        nC = 256 ** 3
    r = bench.roofline_of(Dev(), Grid(), "256V", np.zeros(4, dtype=complex))
    # what the handle did about the placement of its working copies, and the self-check against the committed profile
    assert r["placement"]["tries"] == 3 and abs(r["placement"]["kept_ms"] - 2.3 / 4) < 1e-12 and r["placement"]["reused"]
    if r["rocprof_average"]:
        assert abs(r["vs_profile"] - r["launch_ms"] / r["rocprof_average"]["average_ms"]) < 1e-12
        assert ("placement_mode" in r) == (not 0.97 <= r["vs_profile"] <= 1.03)
    st, sp = r["launch_ms_stats"], r["launch_ms_stats_sparse_source"]
    assert st["samples"] == bench.SWEEP_SAMPLES >= 10 and st["min"] <= st["median"] <= st["max"]
    assert r["launch_ms"] == st["median"] and r["launch_ms_sparse_source"] == sp["median"]
    assert 2.0 <= st["median"] <= 2.1 and 1.0 <= sp["median"] <= 1.05
    assert abs(r["frac"] - 200.0 * 256 ** 3 / 4 / (r["launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-12


def test_max_level_against_the_loop_it_replaces():
    """MGParameters.max_level counts the halvings of a dimension from the trailing zero bits of its cell count (one fewer for a
    pure power of two: 2^k stops at two cells).  Against the plain loop -- halve while even and more than two cells are left
    (the rule of the reference, emg3d/solver.py:1142-1206) -- incl. the user's clevel cap, the coarsest-grid record and the
    "not optimal" note, on powers of two, odd sizes, primes and the smallest grids."""
    from emg3d_amd.solver import MGParameters

    def loop(n):
        k = 0
        while n % 2 == 0 and n > 2:
            k += 1
            n //= 2
        return k, n
    for n in list(range(2, 70)) + [96, 100, 128, 144, 200, 250, 256, 384, 448, 512, 1000, 1024]:
        assert MGParameters._halvings(n) == loop(n)[0], n
    for vnC in [(2, 2, 2), (4, 8, 16), (3, 5, 7), (48, 96, 20), (1024, 2, 6), (128, 128, 128), (100, 40, 24), (144, 200, 36)]:
        for cap in (-1, 0, 1, 2, 3, 5, 9):
            v = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=vnC, clevel=cap)
            d = [min(loop(n)[0], cap) if cap >= 0 else loop(n)[0] for n in vnC]
            assert list(v.pclevel['clevel']) == d
            assert list(v.clevel) == [max(d), max(d[1], d[2]), max(d[0], d[2]), max(d[0], d[1])]
            coarse = tuple(n // 2 ** k for n, k in zip(vnC, d))
            assert v.pclevel['vnC'] == coarse and v.pclevel['nC'] == coarse[0] * coarse[1] * coarse[2]
            lim = np.inf if cap < 0 else cap
            note = any(k < lim and c > 7 for k, c in zip(d, coarse)) or any(k < min(lim, 3) for k in d)
            assert bool(v.pclevel['message']) == note, (vnC, cap)
    with pytest.raises(ValueError, match="at least two"):
        MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=(8, 1, 8))


def test_sweep_plan_thresholds_follow_the_device():
    """The launch selection (which line-sweep kernel family serves a level, at how many lines per wave) is a function of the level's
    shape and of the device's SIMD count, not of literals tuned on one part: `emg3d_sweep_plan` evaluates the
    handle's own predicates on a shape -- no device memory, no launch, no GPU needed with a fake CU count.
    (a) 256 CUs: the selection the GPU tests assert by name (test_gpu_fullsize.py) and DESIGN 3.1 describes.
    (b) 128 CUs: every lines-per-colour threshold halves -- a level with half the lines gets on the small device what the full level
        gets on the large one; rounds of waves double where the launch keeps its shape."""
    from emg3d_amd import _lib
    try:
        _lib.load()
    except _lib.HipLibraryError:
        pytest.skip("library not built")

    def plan(n, d=3, cu=256, **kw):
        return _lib.sweep_plan(n, d, cu_count=cu, **kw)

    # (a) MI355X: 256 CUs = 1024 SIMDs
    assert plan((128,) * 3)["kernel"] == "k_line_sweep_thm<c128,3,8>"            # 4096 lines per colour < 8 x 1024
    p = plan((256,) * 3)
    assert p["kernel"] == "k_line_sweep_qc<c128,2,16>" and p["lines_per_wave"] == 16 and p["rounds"] == 1 and p["factor_kind"] == 4 and p["split"]
    assert plan((144,) * 3)["kernel"] == "k_line_sweep_thm<c128,3,12>"           # 5184 lines: 12 per pair of waves saves a round
    p = plan((200,) * 3)
    assert p["kernel"] == "k_line_sweep_qc<c128,3,16>" and p["lines_per_wave"] == 10 and p["rounds"] == 1
    p = plan((448,) * 3)
    assert p["kernel"] == "k_line_sweep_qc_big<c128,3,16>" and p["big_offsets"] and p["lines_per_wave"] == 13 and p["rounds"] == 4
    p = plan((512,) * 3)
    assert p["kernel"] == "k_line_sweep_qc_big<c128,3,16>" and p["lines_per_wave"] == 16 and p["rounds"] == 4
    assert plan((128, 64, 64), 2)["kernel"] == "k_line_sweep_tha<c128,3>"         # 64-block lines, 2048 lines per colour >= 1100
    assert plan((128, 64, 64), 1)["kernel"] == "k_line_sweep_qpl<c128,4,2>"       # 128-block lines, 1024 lines per colour: the scan kernel
    assert plan((128, 32, 32), 2)["kernel"] == "k_line_sweep_qpl<c128,1,2>"       # 32-block lines: two blocks per quad
    assert plan((128, 8, 8), 3)["kernel"] == "k_line_sweep_qpl<c128,1,1>"         # 8-block lines: scans through LDS
    assert plan((128, 4, 4), 3)["kernel"] == "k_line_sweep_qpl_chain<c128,1,1>"   # 4-block lines: the chain form (one DPP row per line)
    assert plan((128, 4, 4), 3, ordering='lex')["kernel"] == "k_line_sweep_qpl<c128,1,1>"
    assert plan((128,) * 3, dtype=np.float64)["kernel"] == "k_line_sweep_thm<f64,3,8>"
    assert plan((128,) * 3, ordering='lex')["kernel"].startswith("k_line_sweep_qpl<c128,")     # hyperplane launches: the scan kernel
    # level 1 of the 256^3 V-cycle: 8192 lines per colour = 8 lines per wave on every SIMD, three prefetch stages
    p = plan((256, 128, 128), 2)
    assert p["kernel"] == "k_line_sweep_qc<c128,3,16>" and p["lines_per_wave"] == 8 and p["rounds"] == 1

    # (b) a device of half the size: the thresholds are in waves per SIMD
    assert plan((128,) * 3, cu=128)["kernel"].startswith("k_line_sweep_qc<c128,")      # 4096 lines = 8 x 512 SIMDs: the quad kernel's regime
    assert plan((90,) * 3, cu=128)["kernel"] == plan((128,) * 3, cu=256)["kernel"]      # 2025 / 512 ~ 4032 / 1024 lines per SIMD: two-sided
    p = plan((256,) * 3, cu=128)
    assert p["kernel"] == "k_line_sweep_qc<c128,3,16>" and p["lines_per_wave"] == 16 and p["rounds"] == 2
    p = plan((256, 128, 128), 2, cu=128)                                                # 8192 lines = 16 per wave on 512 SIMDs: two stages
    assert p["kernel"] == "k_line_sweep_qc<c128,2,16>" and p["lines_per_wave"] == 16 and p["rounds"] == 1
    # the affine kernel's one-round limit: 8 lines per workgroup, one workgroup per CU (2048 lines at 256 CUs, 1024 at 128)
    assert plan((128, 64, 128), 1, cu=256)["kernel"] == "k_line_sweep_tha<c128,3>"      # 128-block lines, 2048 lines per colour
    assert plan((128, 64, 128), 1, cu=128)["kernel"] != "k_line_sweep_tha<c128,3>"      # two rounds there: the two-sided kernel
    assert plan((128, 64, 64), 1, cu=128)["kernel"] == "k_line_sweep_tha<c128,3>"       # 1024 lines: one round on 128 CUs
    # batched systems: the kernel family and its factor layout do not depend on the batch size (a system stays bit for bit its own
    # solve; lines per pair of waves / prefetch stages -- lane mapping, no arithmetic -- may) ...
    for n in ((128,) * 3, (128, 64, 64), (128, 16, 16), (256,) * 3):
        a, b = plan(n, nsys=8), plan(n)
        assert a["kernel"].split("<")[0] == b["kernel"].split("<")[0] and a["factor_kind"] == b["factor_kind"]
    # ... the rounds do
    assert plan((128,) * 3, nsys=8)["rounds"] > plan((128,) * 3)["rounds"]
