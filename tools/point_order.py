"""Lab experiment: visiting orders of the 8 colours of the point smoother (EMG3D_POINT_ORDER / _B) on problems that use it
(linerelaxation=False).  python tools/point_order.py"""
import itertools, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
from emg3d_amd import _lib

_lib.use(_lib.LAB_PATH)


def problems():
    h = np.ones(32) * 50.
    g = em.TensorMesh([h, h, h], origin=(-800., -800., -800.))
    yield "config 1: 32^3 fullspace F", g, em.Model(g, 1.), em.get_source_field(g, [0, 0, 0, 30, 10], 1.0), dict(cycle='F')
    h = np.ones(64) * 40.
    g = em.TensorMesh([h, h, h], origin=(-1280., -1280., -1280.))
    rng = np.random.default_rng(7)
    rho = 10 ** rng.uniform(0., 1., g.nC)
    yield "64^3 uniform grid, random rho, V", g, em.Model(g, rho), em.get_source_field(g, [0, 0, 0, 30, 10], 2.0), dict(cycle='V')
    h = [em.meshes.stretched_widths(32, 16, 50., f) for f in (1.03, 1.03, 1.03)]
    g = em.TensorMesh(h, origin=[-x.sum() / 2 for x in h])
    yield "64^3 mildly stretched, F + semicoarsening", g, em.Model(g, 1., 2., 3.), em.get_source_field(g, [0, 0, 0, 30, 10], 1.0), dict(cycle='F', semicoarsening=True)


rng = np.random.default_rng(0)
cands = ["01234567", "01234567:01234567", "07254361", "07254361:07254361", "03562174:03562174", "06351742:06351742"]
if len(sys.argv) > 1 and sys.argv[1] == "rotated":
    # even sweeps P, odd sweeps P rotated by one: the odd sweep ends with the colour the even one starts with (skipped):
    # 15 passes per two sweeps, like the reversed pair
    cands = ["01234567", "01234567:01234567"]
    for p in ["01234567", "07254361", "03562174", "06351742", "57631240", "32176054"] + ["".join(str(x) for x in rng.permutation(8)) for _ in range(12)]:
        cands.append(p + ":" + p[1:] + p[0])
else:
    for _ in range(30):
        p = "".join(str(x) for x in rng.permutation(8))
        cands += [p, p + ":" + p]
table = {}
for name, grid, model, sfield, kw in problems():
    print(name, flush=True)
    for c in cands:
        os.environ["EMG3D_POINT_ORDER"] = c.split(":")[0]
        os.environ.pop("EMG3D_POINT_ORDER_B", None)
        if ":" in c:
            os.environ["EMG3D_POINT_ORDER_B"] = c.split(":")[1]
        e, info = em.solve(grid, model, sfield, linerelaxation=False, verb=0, return_info=True, tol=1e-6, maxit=50, **kw)
        err = np.array(info['error_at_cycle']) / info['ref_error']
        rate = (err[-1] / err[1]) ** (1.0 / max(len(err) - 2, 1))
        table.setdefault(c, []).append((info['it_mg'], rate))
print("\nschedule (even sweeps[:odd sweeps as visited; default reversed]): cycles | geometric mean of the reductions")
for c in sorted(cands, key=lambda c: np.mean(np.log([t[1] for t in table[c]]))):
    print(f"{c:18s}: {[t[0] for t in table[c]]}  {float(np.exp(np.mean(np.log([t[1] for t in table[c]])))):.3f}")
