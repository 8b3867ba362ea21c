"""Does the placement effect exist on a level of the size of level 1 of the 256^3 V-cycle (256 x 128 x 128: working copies of 201 MB,
8192 lines per colour)?  Lab library, EMG3D_PLACE_MIN_MB=100: the search runs on level 0 of such a grid and logs its candidates."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("EMG3D_HIP_LIB", os.path.join(os.getcwd(), "emg3d_amd", "libemg3d_hip_lab.so"))
os.environ["EMG3D_PLACE_MIN_MB"] = "100"
os.environ["EMG3D_LOG_SETUP"] = "1"
os.environ.setdefault("EMG3D_PLACE_TRIES", "12")
import numpy as np
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
shape = tuple(int(x) for x in (sys.argv[1:4] or (256, 128, 128)))
h = [em.meshes.stretched_widths(n // 2, n // 4, 25., 1.04) for n in shape]
grid = em.TensorMesh(h, origin=[-hh.sum() / 2 for hh in h])
rho = np.full(grid.nC, 1.0)
model = em.Model(grid, rho, 2 * rho, 3 * rho)
sfield = em.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle='V', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')
with DeviceMG(grid, vm, np.complex128) as dev:
    dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None)
    for sc, lr in zip([1, 2, 3], [4, 5, 6]):
        dev.prepare(sc, lr)
    print(shape, dev.placement())
    print({d: [round(dev.time_sweep(d, 2), 4) for _ in range(3)] for d in (1, 2, 3)}, dev.last_sweep_kernel())
