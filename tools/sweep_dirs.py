"""Isolated level-0 sweep time per direction on an n0 x n1 x n2 stretched grid (ms per sweep of 4 colour launches):
shows how the sweep kernels depend on the memory stride of the line direction."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 64)
freq = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0       # < 0: Laplace domain, float64 kernels
rng = np.random.default_rng(0)
h = [rng.uniform(40, 60, n) for n in shape]
grid = em.TensorMesh(h, origin=(0, 0, 0))
model = em.Model(grid, 1., 2., 3.)
sf = em.get_source_field(grid, [h[0].sum() / 2, h[1].sum() / 2, h[2].sum() / 2, 10, 5], freq)
with DeviceMG(grid, em.VolumeModel(grid, model, sf), sf.dtype) as dev:
    dev.set_params(MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC))
    dev.set_sfield(sf); dev.set_efield(None)
    dev.time_residual(600)
    cyc = dev.cycles(6, [1, 2, 3], [4, 5, 6]) if min(shape) >= 8 else None
    import time
    t0 = time.perf_counter(); dev.cycles(6, [1, 2, 3], [4, 5, 6]); tc = (time.perf_counter() - t0) / 6 * 1e3
    print(shape, sf.dtype, os.environ.get("EMG3D_QPL", ""), f"F-cycle {tc:.2f} ms", {d: round(dev.time_sweep(d, 20), 4) for d in (1, 2, 3)})
