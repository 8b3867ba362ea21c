"""A few isolated sweeps of one direction on an n0 x n1 x n2 grid and nothing else (for rocprofv3 --pmc runs, which
serialise every launch): python tools/sweep_once.py n0 n1 n2 direction [reps]
SWEEP_ONCE_COARSE=1: the conditions of a COARSE level on a level-0 grid -- zeta is read (a model with mu_r: it is not the
cell volume then) and the right-hand side is dense."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
shape = tuple(int(v) for v in sys.argv[1:4])
d = int(sys.argv[4]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rng = np.random.default_rng(0)
h = [rng.uniform(40, 60, n) for n in shape]
grid = em.TensorMesh(h, origin=(0, 0, 0))
coarse = os.environ.get("SWEEP_ONCE_COARSE") == "1"
model = em.Model(grid, 1., 2., 3., mu_r=rng.uniform(1., 2., shape) if coarse else None)
sf = em.get_source_field(grid, [h[0].sum() / 2, h[1].sum() / 2, h[2].sum() / 2, 10, 5], 1.0)
if coarse:
    sf = em.SourceField(grid, (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-9, freq=1.0)
with DeviceMG(grid, em.VolumeModel(grid, model, sf), sf.dtype) as dev:
    dev.set_params(MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC))
    dev.set_sfield(sf); dev.set_efield(None)
    print(shape, d, round(dev.time_sweep(d, reps), 4), dev.last_sweep_kernel())
