"""Localise the fuzz finding: field after ONE cycle with semicoarsening=True (sc_dir 1: level 0 is the coarsest level)
and after the second (sc_dir 2), against the oracle."""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emg3d_amd as em
from oracle import oracle as orc
orc.build()
rng = np.random.default_rng(3)
shape = (8, 3, 3)
h = [rng.uniform(20, 60) * 1.1 ** np.abs(np.arange(n) - n / 2 + 0.5) for n in shape]
grid = em.TensorMesh(h, origin=tuple(-hh.sum() / 2 for hh in h))
rho = 10 ** rng.uniform(-0.5, 2.0, grid.nC)
model = em.Model(grid, rho, property_z=rho * 2)
sfield = em.get_source_field(grid, [1., 2., 0.5, 30., 10.], -0.5)
vm = em.VolumeModel(grid, model, sfield)
om, ov = orc.Mesh(grid.h, grid.origin), orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)
for ordering in ('lex', 'colour'):
    for sc, maxit in ((True, 1), (True, 2), (1, 1), (1, 2), (2, 2), (12, 2), (21, 2), (13, 2), (31, 2)):
        opts = dict(cycle='F', semicoarsening=sc, linerelaxation=1, nu_init=0, nu_pre=2, nu_coarse=2, nu_post=2, maxit=maxit, tol=1e-14)
        e, info = em.solve(grid, model, sfield, return_info=True, verb=0, ordering=ordering, **opts)
        oe, oinfo = orc.solve(om, ov, np.array(sfield), order=0 if ordering == 'lex' else 1, **opts)
        print(ordering, "sc", sc, "maxit", maxit, "field err", np.abs(np.array(e) - oe).max() / np.abs(oe).max(), info['error_at_cycle'], oinfo['error_at_cycle'])
