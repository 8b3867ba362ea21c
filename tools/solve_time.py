"""End-to-end `solve()` wall time on the 128^3 bench problem (set-up + factorisation + cycles + download)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
import emg3d_amd as em
grid, model, sfield, cycle = bench.build_problem(em, sys.argv[1] if len(sys.argv) > 1 else "128F", 1.0)
for k in range(3):
    t0 = time.perf_counter()
    e, info = em.solve(grid, model, sfield, return_info=True, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
    print(f"solve #{k}: {time.perf_counter() - t0:.3f} s, {info['it_mg']} cycles, rel. error {info['rel_error']:.2e}, {info['exit_message']}")
