"""Randomised parity stress of the line/point smoothers on the GPU against the oracle: random grid shapes
(2..70 cells per axis), random widths / models, all four smoother directions, both orderings, nu 1..3.
Not part of the test suite (takes a few minutes); run through gpurun:  python tests/tools/stress.py [ncases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emg3d_amd as em
from oracle import oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
for case in range(ncases):
    vnC = tuple(int(x) for x in rng.choice([2, 3, 4, 5, 6, 8, 9, 12, 16, 17, 24, 31, 32, 33, 40, 48, 64, 65, 70], 3))
    if np.prod(vnC) > 60000:
        vnC = (vnC[0], min(vnC[1], 12), min(vnC[2], 10))
    cplx = rng.random() < 0.7
    h = [rng.uniform(5, 90, n) * rng.choice([1, 1, 3]) for n in vnC]
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    if cplx:
        eta = [np.asfortranarray(-1j * 8e-6 * vol * 10 ** rng.uniform(-1.5, 0.5, grid.vnC)) for _ in range(3)]
        kw = dict(freq=1.)
        rnd = lambda: rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)
    else:
        eta = [np.asfortranarray(-8e-6 * vol * 10 ** rng.uniform(-1.5, 0.5, grid.vnC)) for _ in range(3)]
        kw = dict(freq=-1.)
        rnd = lambda: rng.standard_normal(grid.nE)
    if rng.random() < 0.3:
        eta[1] = eta[0]
        eta[2] = eta[0]
    zeta = np.asfortranarray(vol / rng.uniform(0.9, 1.3, grid.vnC))
    e0 = em.Field(grid, rnd(), **kw); e0.ensure_pec
    s = em.Field(grid, 1e-6 * rnd(), **kw); s.ensure_pec
    nu = int(rng.integers(1, 4))
    for direction in (0, 1, 2, 3):
        for order in (0, 1):
            e = e0.copy()
            em.core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, *eta, zeta, *grid.h, nu, order=order)
            eo = np.array(e0)
            oracle.gauss_seidel(grid.vnC, eo, np.array(s), *eta, zeta, *grid.h, nu, direction=direction, order=order)
            err = np.linalg.norm(np.asarray(e) - eo) / np.linalg.norm(eo)
            worst = max(worst, err)
            if err > 1e-11:
                print("  note", vnC, 'dir', direction, 'order', order, 'nu', nu, f"{err:.2e}", flush=True)
            if not err < 2e-10:
                print("FAIL", vnC, 'complex' if cplx else 'real', 'dir', direction, 'order', order, 'nu', nu, err, flush=True)
    print(case, vnC, 'c128' if cplx else 'f64', f"worst so far {worst:.2e}", flush=True)
print("WORST", worst)
assert worst < 2e-10
