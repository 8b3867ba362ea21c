"""Tiny grids with BiCGSTAB + multigrid preconditioner: how often do iteration counts differ from the oracle?"""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["EMG3D_QPL"] = "0"; os.environ["EMG3D_SPLIT"] = "1"
import emg3d_amd as em
from oracle import oracle as orc
orc.build()
diff = 0
for seed in range(60):
    rng = np.random.default_rng(seed)
    shape = (2, 4, 6)
    h = [rng.uniform(20, 60) * rng.uniform(1.0, 1.3) ** np.abs(np.arange(n) - n / 2 + 0.5) for n in shape]
    grid = em.TensorMesh(h, origin=tuple(-hh.sum() / 2 for hh in h))
    rho = 10 ** rng.uniform(-0.5, 2.0, grid.nC)
    model = em.Model(grid, rho, property_y=rho * rng.uniform(1, 3))
    sfield = em.get_source_field(grid, [rng.uniform(-5, 5), rng.uniform(-10, 10), rng.uniform(-20, 20), rng.uniform(0, 360), rng.uniform(-90, 90)], -0.5)
    vm = em.VolumeModel(grid, model, sfield)
    opts = dict(cycle='V', semicoarsening=False, linerelaxation=True, nu_init=0, nu_pre=0, nu_coarse=1, nu_post=1, maxit=4, tol=1e-12, sslsolver='bicgstab')
    e, info = em.solve(grid, model, sfield, return_info=True, verb=0, ordering='colour', **opts)
    oe, oinfo = orc.solve(orc.Mesh(grid.h, grid.origin), orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield), order=1, **opts)
    same = info['it_mg'] == oinfo['it_mg'] and info['it_ssl'] == oinfo['it_ssl']
    diff += (not same)
    if not same or seed < 3:
        print(seed, "gpu", info['it_mg'], info['it_ssl'], info['exit_message'], info['error_at_cycle'], "| oracle", oinfo['it_mg'], oinfo['it_ssl'], oinfo['exit_message'], oinfo['error_at_cycle'])
print("different iteration counts:", diff, "of 60")
