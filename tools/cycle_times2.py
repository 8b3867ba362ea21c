"""As cycle_times.py with ONE persistent handle (solve(handle=...)): is the stalled cycle tied to handle churn?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.environ.get("NO_TORCH"):
    import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd import models
from emg3d_amd.solver import DeviceMG
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
kw = dict(return_info=True, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
orig = DeviceMG.cycle
times = []
def timed(self, sc, lr):
    t0 = time.perf_counter(); v = orig(self, sc, lr); times.append((time.perf_counter() - t0) * 1e3); return v
DeviceMG.cycle = timed
mode = sys.argv[1] if len(sys.argv) > 1 else "persistent"
parts = models.eta_factored(grid, model, sfield)
dev = DeviceMG.from_sigma_volume(grid, *parts[:4], smu0=parts[4])
for rep in range(4):
    times.clear()
    if mode == "sleep":
        time.sleep(0.05)
    t0 = time.perf_counter()
    em.solve(grid, None, sfield, handle=dev, **kw)
    print(f"{mode}: solve {time.perf_counter() - t0:.3f} s; cycle calls (ms):", " ".join(f"{t:.1f}" for t in times))
