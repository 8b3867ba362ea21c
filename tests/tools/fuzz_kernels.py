"""Randomised kernel-level parity against the oracle (developer aid): amat_x, the four smoothers in both orderings,
restriction and prolongation for every sc_dir the shape allows, get_h_field -- on random shapes (2 ... 40 cells per axis:
2-cell axes, odd sizes), complex128 and float64, aliased and distinct eta, with and without PEC-clean inputs.
    python tests/tools/fuzz_kernels.py [n_cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emg3d_amd as em                      # noqa: E402
from oracle import oracle as orc            # noqa: E402

orc.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
fails = 0


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


for case in range(n_cases):
    if case and case % 500 == 0:
        print(f"... {case} cases so far, {fails} failures", flush=True)
    shape = [int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 17, 20, 24, 32, 40])) for _ in range(3)]
    while np.prod(shape) > 20000:
        shape[int(np.argmax(shape))] = max(shape[int(np.argmax(shape))] // 2, 2)
    cplx = bool(rng.random() < 0.6)
    h = [rng.uniform(10, 80, n) for n in shape]
    grid = em.TensorMesh(h, origin=tuple(rng.uniform(-100, 100, 3)))
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    case_m = int(rng.integers(0, 4))
    smu0 = -1j * 8e-6 if cplx else 3e-6
    base = np.asfortranarray(smu0 * vol * 10 ** rng.uniform(-1.5, 1.0, grid.vnC))
    eta = [base,
           np.asfortranarray(base * rng.uniform(1, 3)) if case_m in (1, 3) else base,
           np.asfortranarray(base * rng.uniform(1, 3)) if case_m in (2, 3) else base]
    zeta = np.asfortranarray(vol / rng.uniform(0.9, 1.3, grid.vnC)) if rng.random() < 0.5 else np.asfortranarray(vol.copy())
    freq = 1.0 if cplx else -1.0

    def rnd(g, pec=True, scale=1.0):
        v = rng.standard_normal(g.nE) + (1j * rng.standard_normal(g.nE) if cplx else 0)
        f = em.Field(g, (scale * v).astype(np.complex128 if cplx else np.float64), freq=freq)
        if pec:
            f.ensure_pec
        return f

    e0, s = rnd(grid), rnd(grid, scale=1e-6)
    errs = {}
    # amat_x
    r = s.copy(); ro = np.array(s)
    em.core.amat_x(r.fx, r.fy, r.fz, e0.fx, e0.fy, e0.fz, *eta, zeta, *grid.h)
    orc.amat_x(grid.vnC, ro, np.array(e0), *eta, zeta, *grid.h)
    errs['amat_x'] = rel(r, ro)
    # smoothers
    for direction in range(4):
        for order in (0, 1):
            nu = int(rng.integers(1, 4))
            e = e0.copy(); eo = np.array(e0)
            em.core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, *eta, zeta, *grid.h, nu, order=order)
            orc.gauss_seidel(grid.vnC, eo, np.array(s), *eta, zeta, *grid.h, nu, direction=direction, order=order)
            errs[f'gs{direction}o{order}'] = rel(e, eo)

    class VM:
        eta_x, eta_y, eta_z = eta
        case = case_m
    VM.zeta = zeta
    om = orc.Mesh(grid.h, grid.origin)
    ov = orc.VModel(eta[0], eta[1], eta[2], zeta, case_m)
    res = rnd(grid, pec=False)
    for sc_dir in range(7):
        co = [sc_dir not in sk for sk in ([1, 5, 6], [2, 4, 6], [3, 4, 5])]
        if any(c and (n % 2 or n < 4) for c, n in zip(co, shape)) and any(c and n % 2 for c, n in zip(co, shape)):
            continue                        # an odd axis cannot be coarsened
        if any(c and n < 2 for c, n in zip(co, shape)):
            continue
        cgrid, cmodel, cs, ce = em.solver.restriction(grid, VM, s, res, sc_dir)
        ocm, ocmod, ocs, oce = orc.restriction(om, ov, np.array(s), np.array(res), sc_dir)
        errs[f'restrict{sc_dir}'] = max(rel(cs, ocs), rel(cmodel.eta_x, ocmod.eta_x), rel(cmodel.eta_z, ocmod.eta_z),
                                        rel(cmodel.zeta, ocmod.zeta))
        cev = rnd(cgrid, pec=False)
        e = e0.copy(); eo = np.array(e0)
        em.solver.prolongation(grid, e, cgrid, cev, sc_dir)
        orc.prolongation(om, eo, ocm, np.array(cev), sc_dir)
        errs[f'prolong{sc_dir}'] = rel(e, eo)
    # magnetic field
    model = em.Model(grid, 10 ** rng.uniform(-1, 1, grid.nC), mu_r=rng.uniform(0.8, 1.5, grid.nC) if rng.random() < 0.5 else None)
    hf = em.get_h_field(grid, model, e0)
    zh = None if model.mu_r is None else np.asfortranarray(vol / model.mu_r)
    errs['hfield'] = rel(np.asarray(hf), orc.get_h_field(om, np.array(e0), e0.smu0, zh))
    bad = {k: v for k, v in errs.items() if not v < 2e-10}
    for k, v in errs.items():
        worst[k[:8]] = max(worst.get(k[:8], 0.0), v)
    if bad:
        fails += 1
        print(f"{case:3d} {tuple(shape)} cplx={cplx} case={case_m}: FAIL {bad}", flush=True)
print(f"{n_cases} cases, {fails} failures; worst: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items())))
sys.exit(1 if fails else 0)
