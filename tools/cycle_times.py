"""Wall time of every DeviceMG.cycle call inside solve() (128^3 bench problem): set-up of the three rotation
states happens inside the first three calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
kw = dict(return_info=True, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
orig = DeviceMG.cycle
times = []
def timed(self, sc, lr):
    t0 = time.perf_counter(); v = orig(self, sc, lr); times.append((time.perf_counter() - t0) * 1e3); return v
DeviceMG.cycle = timed
for rep in range(3):
    times.clear()
    t0 = time.perf_counter()
    em.solve(grid, model, sfield, **kw)
    print(f"solve {time.perf_counter() - t0:.3f} s; cycle calls (ms):", " ".join(f"{t:.1f}" for t in times))
