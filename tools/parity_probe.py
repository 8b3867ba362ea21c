import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_golden
import emg3d_amd as em
g = load_golden("regression.npz")
def reg(pre):
    grid = em.TensorMesh([g[f'{pre}_hx'], g[f'{pre}_hy'], g[f'{pre}_hz']], origin=g[f'{pre}_origin'])
    model = em.Model(grid, g[f'{pre}_property_x'], g[f'{pre}_property_y'], g[f'{pre}_property_z'])
    sfield = em.SourceField(grid, g[f'{pre}_sfield'].copy(), freq=float(g[f'{pre}_freq']))
    return grid, model, sfield
for pre, key, kw in [('res','F',{}),('res','W',{'cycle':'W'}),('res','V',{'cycle':'V'}),('res','bic',{'sslsolver':True}),('lap','F',{}),('lap','bic',{'sslsolver':True})]:
    grid, model, sfield = reg(pre)
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    ref = g[f'{pre}_{key}_error_at_cycle']; got = info['error_at_cycle']
    dev = np.abs(got-ref)/ref
    print(pre, key, 'cycles', len(ref)-1, 'norm dev per cycle', ' '.join(f'{d:.1e}' for d in dev), '| field rel err', f"{np.linalg.norm(np.asarray(e)-g[f'{pre}_{key}_here'])/np.linalg.norm(g[f'{pre}_{key}_here']):.1e}")
gs = load_golden("solves_16.npz")
grid = em.TensorMesh([gs['hx'], gs['hy'], gs['hz']], origin=gs['origin'])
model = em.Model(grid, gs['rho_b'], 2*gs['rho_b'], 3*gs['rho_b'])
sfield = em.get_source_field(grid, gs['src'], float(gs['freq']))
for name, kw in [('F_sclr', dict(cycle='F', semicoarsening=True, linerelaxation=True)), ('V_sclr', dict(cycle='V', semicoarsening=True, linerelaxation=True))]:
    e, info = em.solve(grid, model, sfield, return_info=True, ordering='lex', **kw)
    ref = gs[f'{name}_error_at_cycle']; got = info['error_at_cycle']
    dev = np.abs(got-ref)/ref
    print('s16', name, 'norm dev per cycle', ' '.join(f'{d:.1e}' for d in dev), '| field', f"{np.linalg.norm(np.asarray(e)-gs[f'{name}_efield'])/np.linalg.norm(gs[f'{name}_efield']):.1e}")
