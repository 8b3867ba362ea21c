"""Randomised bit-identity sweep of the residual kernel variants (block maps x node planes per thread x batch) against
the plain kernel: python tests/tools/fuzz_residual.py [cases] [seed].  GPU; one-off tool, not collected by pytest."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ["EMG3D_RES_ZM_MIN_CELLS"] = "0"
os.environ["EMG3D_RES_XCD_MIN"] = "1"
bad = 0
for case in range(ncase):
    shape = tuple(int(x) for x in rng.integers(2, 70, 3))
    if np.prod(shape) > 150000:
        continue
    cplx = rng.random() < 0.7
    dtype = np.complex128 if cplx else np.float64
    freq = 1.3 if cplx else -2.0
    grid = em.TensorMesh([rng.uniform(5., 50., n) for n in shape], origin=(0., 0., 0.))
    model = em.Model(grid, *(10 ** rng.uniform(-1, 2, shape) for _ in range(3)),
                     mu_r=rng.uniform(1., 3., shape) if rng.random() < 0.5 else None)
    sf = em.SourceField(grid, freq=freq)
    vm = em.VolumeModel(grid, model, sf)
    nsys = int(rng.integers(1, 4))
    s = [(rng.standard_normal(grid.nE) + (1j * rng.standard_normal(grid.nE) if cplx else 0)).astype(dtype) for _ in range(nsys)]
    e = [(rng.standard_normal(grid.nE) + (1j * rng.standard_normal(grid.nE) if cplx else 0)).astype(dtype) for _ in range(nsys)]
    ref = None
    for xcd, kz in [("0", "1")] + [(str(rng.integers(0, 3)), str(2 ** rng.integers(0, 5))) for _ in range(4)]:
        os.environ["EMG3D_RES_XCD"] = xcd
        os.environ["EMG3D_RES_KZ"] = kz
        with DeviceMG(grid, vm, np.dtype(dtype)) as dev:
            if nsys > 1:
                dev.set_batch(nsys)
            for b in range(nsys):
                dev.select(b)
                dev.set_sfield(em.Field(grid, s[b], freq=freq))
                dev.set_efield(em.Field(grid, e[b], freq=freq))
            norms = np.atleast_1d(dev.residual_norm())
            res = []
            for b in range(nsys):
                dev.select(b)
                res.append(dev.get_residual())
            name = dev.last_residual_kernel()
        if ref is None:
            ref = (norms, res)
        else:
            ok = np.array_equal(norms, ref[0]) and all(np.array_equal(a, b) for a, b in zip(res, ref[1]))
            if not ok:
                bad += 1
                print("MISMATCH", shape, dtype.__name__, nsys, xcd, kz, name, flush=True)
print(f"{ncase} cases, {bad} mismatches", flush=True)
