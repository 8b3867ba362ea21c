"""In-kernel timestamps of k_line_sweep_tha (lab build, EMG3D_Q_TILE=256): one smoothing call along y on 128 x 64 x 64."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 64, 64)
d = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rng = np.random.default_rng(0)
h = [rng.uniform(40, 60, n) for n in shape]
grid = em.TensorMesh(h, origin=(0, 0, 0))
model = em.Model(grid, 1., 2., 3.)
sf = em.get_source_field(grid, [h[0].sum() / 2, h[1].sum() / 2, h[2].sum() / 2, 10, 5], 1.0)
with DeviceMG(grid, em.VolumeModel(grid, model, sf), sf.dtype) as dev:
    dev.set_params(MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC))
    dev.set_sfield(sf); dev.set_efield(None)
    dev.smooth(1, d)
    print("warm", flush=True)
    dev.smooth(1, d)
    print(dev.last_sweep_kernel())
