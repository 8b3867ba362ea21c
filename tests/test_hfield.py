"""Magnetic field H = -curl E / (s mu_0) (reference emg3d/fields.py:819-911).

CPU: the oracle's restatement against tests/golden/hfield.npz (captured by importing the reference,
tests/golden/make_hfield_golden.py) -- bit for bit -- and against the reference's own stored golden
`reg_2>hresult` (tests/test_fields.py:351-362, assert_allclose i.e. rtol 1e-7).
GPU: the HIP kernel k_hfield through the C ABI (stateless emg3d_get_h_field and the handle's
emg3d_mg_get_hfield) against the same vectors and the oracle."""
import numpy as np
import pytest
from scipy.constants import mu_0

from conftest import load_golden, relerr

# complex128 / float64 curl of O(1) values; the kernel follows NumPy's division arithmetic
# (multiply by the reciprocal for complex / real) with contraction off: agreement to the last bits
TOL = 2e-15
CASES = [('reg2', 'reg2_h_here', None), ('res', 'res_h_nomur', None), ('res', 'res_h_mur1', 1.),
         ('res', 'res_h_mur2', 2.), ('mur_c128', 'mur_c128_h', 'array'), ('mur_f64', 'mur_f64_h', 'array')]


def _case(g, pre, mur):
    h = [g[f'{pre}_h{c}'] for c in 'xyz']
    f = float(g[f'{pre}_freq'])
    smu0 = np.array(-2j * np.pi * f) * mu_0 if f > 0 else np.array(f) * mu_0
    mu_r = g[f'{pre}_mu_r'] if isinstance(mur, str) else mur
    return h, f, smu0, mu_r


@pytest.mark.parametrize("pre,key,mur", CASES)
def test_oracle_matches_reference(oracle, pre, key, mur):
    g = load_golden("hfield.npz")
    h, f, smu0, mu_r = _case(g, pre, mur)
    m = oracle.Mesh(h, [0, 0, 0])
    zeta = None if mu_r is None else m.cell_volumes / (np.reshape(mu_r, m.vnC, order='F') if np.ndim(mu_r) else mu_r)
    out = oracle.get_h_field(m, g[f'{pre}_e'], smu0, zeta)
    assert np.array_equal(out, g[key])
    if pre == 'reg2':
        # the reference's stored vector predates CODATA-2018 (mu_0 moved by 5.5e-10); its own test uses rtol 1e-7
        np.testing.assert_allclose(out, g['reg2_h_golden'], rtol=1e-7, atol=0)
        assert abs(smu0 - g['reg2_smu0']) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pre,key,mur", CASES)
def test_hip_matches_reference(pre, key, mur):
    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG
    g = load_golden("hfield.npz")
    h, f, smu0, mu_r = _case(g, pre, mur)
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    model = em.Model(grid, 1., mu_r=mu_r)
    e = em.Field(grid, g[f'{pre}_e'].copy(), freq=f)
    hf = em.get_h_field(grid, model, e)
    assert hf.is_electric is False and hf.dtype == g[key].dtype and hf.freq is None
    nx, ny, nz = grid.vnC
    assert hf.fx.shape == (nx + 1, ny, nz) and hf.fy.shape == (nx, ny + 1, nz) and hf.fz.shape == (nx, ny, nz + 1)
    assert relerr(hf, g[key]) < TOL
    if pre == 'reg2':
        np.testing.assert_allclose(hf, g['reg2_h_golden'], rtol=1e-7, atol=0)
    # device-resident field of a handle: same kernel, only H crosses PCIe
    sf = em.SourceField(grid, freq=f)
    with DeviceMG(grid, em.VolumeModel(grid, model, sf), e.dtype) as dev:
        dev.set_efield(e)
        h2 = dev.get_hfield(grid, smu0, mu_r=mu_r is not None)
    assert np.array_equal(np.asarray(h2), np.asarray(hf))


@pytest.mark.gpu
def test_hip_mu_r_one_is_neutral_and_errors():
    """Reference tests/test_fields.py:366-381: mu_r = 1 gives the field of no mu_r (to rounding), mu_r = 2 not."""
    import emg3d_amd as em
    g = load_golden("hfield.npz")
    h, f, smu0, _ = _case(g, 'res', None)
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    e = em.Field(grid, g['res_e'].copy(), freq=f)
    h0 = em.get_h_field(grid, em.Model(grid, 1.), e)
    h1 = em.get_h_field(grid, em.Model(grid, 1., mu_r=1.), e)
    h2 = em.get_h_field(grid, em.Model(grid, 1., mu_r=2.), e)
    np.testing.assert_allclose(h0, h1)
    assert not np.allclose(h0, h2)
    with pytest.raises(ValueError):
        em.get_h_field(grid, em.Model(grid, 1.), em.Field(grid))        # no frequency
    from emg3d_amd import _lib
    lib = _lib.load()
    assert lib.emg3d_get_h_field(1, 0, 4, 4, None, None, None, None, None, None, 0.0, 1.0) == -2
    out = np.zeros(10)
    assert lib.emg3d_get_h_field(0, 2, 2, 2, _lib.ptr(out), _lib.ptr(out), None, _lib.ptr(out), _lib.ptr(out),
                                 _lib.ptr(out), 0.0, 0.0) == -2          # smu0 = 0


@pytest.mark.gpu
@pytest.mark.parametrize("vnC", [(128, 128, 128), (2, 3, 130), (257, 2, 2)])
def test_hip_curl_of_gradient_vanishes_fullsize(vnC):
    """Size-independent property at the benchmark size: the discrete curl of a discrete gradient is zero
    (to rounding of the differences), with and without mu_r; and H is linear in E."""
    import emg3d_amd as em
    rng = np.random.default_rng(3)
    h = [rng.uniform(0.5, 2., n) for n in vnC]
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    phi = rng.standard_normal([n + 1 for n in vnC]) + 1j * rng.standard_normal([n + 1 for n in vnC])
    ex = np.diff(phi, axis=0) / h[0][:, None, None]
    ey = np.diff(phi, axis=1) / h[1][None, :, None]
    ez = np.diff(phi, axis=2) / h[2][None, None, :]
    e = em.Field(ex, ey, ez, freq=1.0)
    scale = np.abs(phi).max() / (min(hh.min() for hh in h) ** 2 * abs(e.smu0))
    for model in (em.Model(grid, 1.), em.Model(grid, 1., mu_r=rng.uniform(0.5, 2., grid.nC))):
        hf = em.get_h_field(grid, model, e)
        assert hf.size == sum((vnC[0] + (c == 0)) * (vnC[1] + (c == 1)) * (vnC[2] + (c == 2)) for c in range(3))
        assert np.abs(hf).max() < 1e-13 * scale
    e2 = em.Field(grid, rng.standard_normal(grid.nE) + 0j, freq=1.0)
    m = em.Model(grid, 1.)
    lhs = em.get_h_field(grid, m, em.Field(grid, np.asarray(e) + 2 * np.asarray(e2), freq=1.0))
    assert relerr(lhs, 2 * np.asarray(em.get_h_field(grid, m, e2))) < 1e-12
