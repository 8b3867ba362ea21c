import sys, numpy as np
sys.path.insert(0,'.')
import emg3d_amd as em
g=np.load('tests/golden/solves_16.npz')
grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
sfield = em.get_source_field(grid, g['src'], float(g['freq']))
e, info = em.solve(grid, model, sfield, return_info=True, ordering='colour', cycle='F', semicoarsening=True, linerelaxation=True)
print(info['it_mg'], info['exit_message'], info['error_at_cycle'][:4])
