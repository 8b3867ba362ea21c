"""Randomised check of re-targeted handles (emg3d_mg_set_smu0): shard.solve_frequencies (one handle per dtype for the whole
frequency list) against em.solve() with a handle of its own per frequency, and shard.solve_survey against
solver.solve_sources -- fields, histories and responses bit for bit.  python tests/tools/fuzz_reuse.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import emg3d_amd as em
from emg3d_amd import shard, solver

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
SIZES = [2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24]
fails = 0
for case in range(ncase):
    rng = np.random.default_rng([seed, case])
    while True:
        shape = [int(rng.choice(SIZES)) for _ in range(3)]
        if np.prod(shape) <= 6000 and max(shape) >= 4 and min(shape) >= 3:     # (receivers need >= 3 cells per axis)
            break
    h = [rng.uniform(20, 60) * rng.uniform(1.0, 1.3) ** np.abs(np.arange(n) - n / 2 + 0.5) for n in shape]
    grid = em.TensorMesh(h, origin=tuple(-hh.sum() / 2 for hh in h))
    rho = 10 ** rng.uniform(-0.5, 2.0, grid.vnC)
    kw = {}
    aniso = int(rng.integers(0, 4))
    if aniso in (1, 3):
        kw['property_y'] = rho * rng.uniform(1, 3)
    if aniso in (2, 3):
        kw['property_z'] = rho * rng.uniform(1, 3)
    if rng.random() < 0.3:
        kw['mu_r'] = rng.uniform(1, 2, grid.vnC)
    model = em.Model(grid, rho, **kw)
    opts = dict(cycle=str(rng.choice(['V', 'W', 'F'])), semicoarsening=rng.choice([False, True, 1, 2, 3, 12, 231]).item(),
                linerelaxation=rng.choice([False, True, 4, 5, 6, 7, 1, 2, 3, 56]).item(), verb=0, maxit=int(rng.integers(1, 6)),
                tol=1e-30, ordering=str(rng.choice(['colour', 'lex'])))
    freqs = [float(rng.choice([0.1, 0.5, 1.0, 3.0, 7.0, -0.5, -2.0, -5.0])) for _ in range(int(rng.integers(2, 6)))]
    ext = [hh.sum() / 4 for hh in h]
    srcs = [[rng.uniform(-e_, e_) for e_ in ext] + [rng.uniform(0, 360), rng.uniform(-90, 90)] for _ in range(3)]
    rec = tuple(np.array([rng.uniform(-e_, e_) for _ in range(3)]) for e_ in ext) + (np.array([0., 30., 90.]), np.array([0., 10., -40.]))
    tag = f"{case} {tuple(shape)} aniso={aniso} mu={'mu_r' in kw} {opts['ordering']} {opts['cycle']} sc={opts['semicoarsening']} lr={opts['linerelaxation']} maxit={opts['maxit']} freqs={freqs}"
    try:
        ok = True
        res = shard.solve_frequencies(grid, model, srcs[0], freqs, rec=rec, **opts)
        for f, (e, info, r) in zip(freqs, res):
            e1, info1 = em.solve(grid, model, em.SourceField(grid, freq=f), source=(srcs[0], 0), return_info=True, **opts)
            r1 = em.get_receiver_response(grid, e1, rec)
            ok = ok and np.array_equal(np.asarray(e), np.asarray(e1)) and np.array_equal(info['error_at_cycle'], info1['error_at_cycle'])
            ok = ok and np.allclose(r, r1, rtol=1e-12, atol=0, equal_nan=True)
        resp, infos, efs = shard.solve_survey(grid, model, srcs, freqs, rec, batch=2, return_fields=True, **opts)
        for jf, f in enumerate(freqs):
            for i0 in (0, 2):
                e, info, r = solver.solve_sources(grid, model, srcs[i0:i0 + 2], f, rec=rec, **opts)
                for k in range(len(e)):
                    ok = ok and np.array_equal(np.asarray(efs[i0 + k][jf]), np.asarray(e[k]))
        print(tag, "ok" if ok else "FAIL", flush=True)
        fails += (not ok)
    except Exception as ex:      # noqa
        print(tag, f"EXCEPTION {type(ex).__name__}: {ex}", flush=True)
        fails += 1
print(f"{ncase} cases, {fails} failures", flush=True)
