"""
ORACLE -- test infrastructure, NOT product code.

CPU restatement (NumPy) of the adjoint-state gradient of ONE (source, frequency) pair on its computational
grid, emg3d v0.17.0:

* ``edges2cellaverages``     reference emg3d/maps.py:578-630 (numba kernel)
* ``residual_strengths``     reference emg3d/simulations.py:1171-1213 (``_get_rfield``: receivers become sources
                             with strength conj(residual) conj(weight) / s mu_0)
* ``misfit`` / ``gradient``  reference emg3d/optimize.py:100-111 and 176-199 (Equation (10) of Plessix & Mulder
                             2008: -Re(lambda E s mu_0), edges -> volume-weighted cell averages, sum of components)

The reference wraps these in ``Simulation`` / ``Survey`` (xarray containers, out of scope here and not importable
in this container); the arithmetic above is what those classes call.  Parity status: PINNED by
tests/golden/gradient.npz (the reference's own functions composed by tests/golden/make_golden.py).
"""
import numpy as np


def edges2cellaverages(ex, ey, ez, vol):
    """reference emg3d/maps.py:578-630: same loop order and statement order (pure Python loops: small grids)."""
    nx, ny, nz = vol.shape
    out_x = np.zeros(vol.shape, dtype=ex.dtype)
    out_y = np.zeros(vol.shape, dtype=ex.dtype)
    out_z = np.zeros(vol.shape, dtype=ex.dtype)
    for iz in range(nz + 1):
        izm, izp = max(0, iz - 1), min(nz - 1, iz)
        for iy in range(ny + 1):
            iym, iyp = max(0, iy - 1), min(ny - 1, iy)
            for ix in range(nx + 1):
                ixm, ixp = max(0, ix - 1), min(nx - 1, ix)
                if ix < nx:
                    out_x[ix, iym, izm] += vol[ix, iym, izm] * ex[ix, iy, iz] / 4
                    out_x[ix, iyp, izm] += vol[ix, iyp, izm] * ex[ix, iy, iz] / 4
                    out_x[ix, iym, izp] += vol[ix, iym, izp] * ex[ix, iy, iz] / 4
                    out_x[ix, iyp, izp] += vol[ix, iyp, izp] * ex[ix, iy, iz] / 4
                if iy < ny:
                    out_y[ixm, iy, izm] += vol[ixm, iy, izm] * ey[ix, iy, iz] / 4
                    out_y[ixp, iy, izm] += vol[ixp, iy, izm] * ey[ix, iy, iz] / 4
                    out_y[ixm, iy, izp] += vol[ixm, iy, izp] * ey[ix, iy, iz] / 4
                    out_y[ixp, iy, izp] += vol[ixp, iy, izp] * ey[ix, iy, iz] / 4
                if iz < nz:
                    out_z[ixm, iym, iz] += vol[ixm, iym, iz] * ez[ix, iy, iz] / 4
                    out_z[ixp, iym, iz] += vol[ixp, iym, iz] * ez[ix, iy, iz] / 4
                    out_z[ixm, iyp, iz] += vol[ixm, iyp, iz] * ez[ix, iy, iz] / 4
                    out_z[ixp, iyp, iz] += vol[ixp, iyp, iz] * ez[ix, iy, iz] / 4
    return out_x, out_y, out_z


def misfit(synthetic, observed, weights):
    """reference emg3d/optimize.py:100-111."""
    r = synthetic - observed
    return float(np.sum(weights * (r.conj() * r)).real / 2), r


def residual_strengths(residual, weights, smu0):
    """Strength of the residual source at every receiver, reference emg3d/simulations.py:1184-1188."""
    return residual.conj() * np.conj(weights) / smu0


def gradient_on_grid(vnC, vol, efield, bfield, smu0):
    """reference emg3d/optimize.py:176-199: -Re(bfield * efield * smu0) on the edges, mapped to volume-weighted cell
    averages, components added.  (The reference then maps -grad to the model grid: maps.grid2grid, out of scope.)"""
    nx, ny, nz = vnC
    shp = ((nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz))
    off = np.cumsum([0] + [int(np.prod(s)) for s in shp])
    prod = -np.real(bfield * efield * smu0)
    comps = [prod[off[c]:off[c + 1]].reshape(shp[c], order='F') for c in range(3)]
    gx, gy, gz = edges2cellaverages(comps[0], comps[1], comps[2], vol)
    return gx + gy + gz
