"""A few isolated sweeps of one direction on an n0 x n1 x n2 grid and nothing else (for rocprofv3 --pmc runs, which
serialise every launch): python tools/sweep_once.py n0 n1 n2 direction [reps]"""
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
model = em.Model(grid, 1., 2., 3.)
sf = em.get_source_field(grid, [h[0].sum() / 2, h[1].sum() / 2, h[2].sum() / 2, 10, 5], 1.0)
with DeviceMG(grid, em.VolumeModel(grid, model, sf), sf.dtype) as dev:
    dev.set_params(MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC))
    dev.set_sfield(sf); dev.set_efield(None)
    print(shape, d, round(dev.time_sweep(d, reps), 4), dev.last_sweep_kernel())
