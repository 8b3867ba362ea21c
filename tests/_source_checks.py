"""The checks of the reference's tests/test_fields.py (test_get_source_field, test_arbitrarily_shaped_source,
test_get_source_field_point_vs_finite) written once for a source builder ``get(grid, src, freq, strength=0, electric=True,
length=1.0)`` returning a SourceField -- run on the host twin (CPU test) and on the device path (GPU test)."""
import numpy as np
import pytest
from scipy import constants


def run(get, meshes, fields):
    src = [100, 200, 300, 27, 31]
    h = np.ones(4)
    grid = meshes.TensorMesh([h * 200, h * 400, h * 800], (-450., -850., -1650.))
    freq = 1.2458
    sfield = get(grid, src, freq, strength=1 + 1j)
    np.testing.assert_array_equal(sfield.strength, complex(1 + 1j))
    for f, smu in ((freq, -2j * np.pi * freq * constants.mu_0), (-freq, -freq * constants.mu_0)):
        sfield = get(grid, src, f, strength=0)
        np.testing.assert_array_equal(sfield.strength, float(0))
        assert 4 == sfield.fx[sfield.fx != 0].size and 4 == sfield.fy[sfield.fy != 0].size and 4 == sfield.fz[sfield.fz != 0].size
        hh = np.cos(np.deg2rad(src[4]))
        y, x, z = np.sin(np.deg2rad(src[3])) * hh, np.cos(np.deg2rad(src[3])) * hh, np.sin(np.deg2rad(src[4]))
        np.testing.assert_allclose(np.sum(sfield.fx / x / (-smu)).real, -1)
        np.testing.assert_allclose(np.sum(sfield.fy / y / (-smu)).real, -1)
        np.testing.assert_allclose(np.sum(sfield.fz / z / (-smu)).real, -1)
        np.testing.assert_allclose(np.sum(sfield.vx / x), 1)
        np.testing.assert_allclose(np.sum(sfield.vy / y), 1)
        np.testing.assert_allclose(np.sum(sfield.vz / z), 1)
        assert sfield._freq == f and sfield.freq == freq
        np.testing.assert_allclose(sfield.smu0, smu)
    # source on the final node
    s6 = [grid.nodes_x[0], grid.nodes_x[0] + 1, grid.nodes_y[-1] - 1, grid.nodes_y[-1], grid.nodes_z[0], grid.nodes_z[0] + 1]
    sfield = get(grid, s6, freq)
    tot = np.linalg.norm([np.sum(sfield.fx), np.sum(sfield.fy), np.sum(sfield.fz)])
    np.testing.assert_allclose(tot / np.abs(2j * np.pi * freq * constants.mu_0), 1.0)
    with pytest.raises(ValueError, match='Source is wrong defined'):
        get(grid, [0, 0, 0, 0], 1)
    with pytest.raises(ValueError, match='Provided source outside grid'):
        get(grid, [1e10, 1e10, 1e10, 0, 0], 1)
    with pytest.raises(ValueError, match='Provided finite dipole has no leng'):
        get(grid, [0, 0, 100, 100, -200, -200], 1)

    # arbitrarily shaped sources and the magnetic point dipole
    grid = meshes.TensorMesh([h * 200, h * 400, h * 800], [-400., -800., -1600.])
    freq, strength, src = 1.11, np.pi, (0, 0, 0, 0, 90)
    with pytest.raises(ValueError, match='All source coordinates must have'):
        get(grid, ([1, 2], 1, 1), freq, strength)
    segs = [np.r_[src[0] - 0.5, src[0] + 0.5, src[1] - 0.5, src[1] - 0.5, src[2], src[2]],
            np.r_[src[0] + 0.5, src[0] + 0.5, src[1] - 0.5, src[1] + 0.5, src[2], src[2]],
            np.r_[src[0] + 0.5, src[0] - 0.5, src[1] + 0.5, src[1] + 0.5, src[2], src[2]],
            np.r_[src[0] - 0.5, src[0] - 0.5, src[1] + 0.5, src[1] - 0.5, src[2], src[2]]]
    path = ([src[0] - 0.5, src[0] + 0.5, src[0] + 0.5, src[0] - 0.5, src[0] - 0.5],
            [src[1] - 0.5, src[1] - 0.5, src[1] + 0.5, src[1] + 0.5, src[1] - 0.5], [src[2]] * 5)
    for st_seg, st_path in ((strength, strength), (0.25, 0)):
        sman = fields.SourceField(grid, freq=freq)
        for srcl in segs:
            sman += get(grid, srcl, freq, st_seg)
        scomp = get(grid, path, freq, st_path)
        np.testing.assert_allclose(np.array(sman), np.array(scomp), rtol=1e-12, atol=1e-18)     # (fields ~1e-6)
    scomp = get(grid, path, freq, strength)
    scomp2 = get(grid, src, freq, strength, electric=False)
    np.testing.assert_allclose(-scomp2.vector, scomp.vector, rtol=1e-6, atol=1e-15)

    # point dipole vs finite dipole
    def f_src(d, slen=1.0):
        hh = np.cos(np.deg2rad(d[4]))
        xyz = [np.cos(np.deg2rad(d[3])) * hh, np.sin(np.deg2rad(d[3])) * hh, np.sin(np.deg2rad(d[4]))]
        return [d[0] - xyz[0] * slen / 2, d[0] + xyz[0] * slen / 2, d[1] - xyz[1] * slen / 2, d[1] + xyz[1] * slen / 2,
                d[2] - xyz[2] * slen / 2, d[2] + xyz[2] * slen / 2]
    h3 = np.ones(3) * 500
    grid1 = meshes.TensorMesh([h3, h3, h3], np.array([-750., -750., -750.]))
    d = [0, 0., 0., 23, 15]
    np.testing.assert_allclose(np.array(get(grid1, f_src(d), 1)), np.array(get(grid1, d, 1)), rtol=1e-9, atol=1e-22)
    d = [0, 0., 0., 32, 53]
    np.testing.assert_allclose(np.array(get(grid1, f_src(d), 3.3, np.pi)), np.array(get(grid1, d, 3.3, np.pi)), rtol=1e-9, atol=1e-22)
    h8 = np.ones(8) * 200
    grid2 = meshes.TensorMesh([h8, h8, h8], np.array([-800., -800., -800.]))
    d = [0, 0., 0., 40, 20]
    dsf, fsf = get(grid2, d, 10.0, 0, length=300.0), get(grid2, f_src(d, 300.0), 10.0, 0)
    for c in ('fx', 'fy', 'fz'):
        np.testing.assert_allclose(getattr(fsf, c).sum(), getattr(dsf, c).sum())
