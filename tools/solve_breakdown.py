"""Where the wall time of ONE solve() goes: python tools/solve_breakdown.py [64F] -- wraps the DeviceMG methods with timers
(warm process: second solve)."""
import os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd import solver, models, fields
from emg3d_amd.solver import DeviceMG

wl = sys.argv[1] if len(sys.argv) > 1 else "64F"
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
acc = collections.OrderedDict()


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    lab = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(lab, []).append(1e3 * (time.perf_counter() - t0))
        return r
    setattr(obj, name, g)


for n in ("set_params", "set_sfield", "set_efield", "begin", "residual_norm", "sfield_norm", "cycle", "get_efield", "close",
          "smooth", "prepare"):
    wrap(DeviceMG, n)
wrap(DeviceMG, "__init__", "DeviceMG()")
_fsv = DeviceMG.from_sigma_volume.__func__
def fsv(cls, *a, **k):
    t0 = time.perf_counter(); r = _fsv(cls, *a, **k); acc.setdefault("from_sigma_volume", []).append(1e3 * (time.perf_counter() - t0)); return r
DeviceMG.from_sigma_volume = classmethod(fsv)
wrap(models, "eta_factored")
for rep in range(3):
    acc.clear()
    t0 = time.perf_counter()
    e, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0, return_info=True)
    tot = 1e3 * (time.perf_counter() - t0)
    print(f"{wl} rep {rep}: solve {tot:.1f} ms, {info['it_mg']} cycles; " +
          ", ".join(f"{k} {sum(v):.1f}" + (f" ({' '.join(f'{x:.1f}' for x in v)})" if len(v) > 1 else "") for k, v in acc.items()) +
          f"; unaccounted {tot - sum(sum(v) for k, v in acc.items() if k not in ('DeviceMG()',)):.1f}", flush=True)
