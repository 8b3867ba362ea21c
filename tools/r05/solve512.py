"""A complete solve() at 512^3 (134 M cells, 403 M unknowns) on one MI355X: model set-up on the host, upload, hierarchy and line
factorisations, multigrid cycles to tol = 1e-6, download.  python tools/r05/solve512.py [workload] [cycle]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import emg3d_amd as em

wl = sys.argv[1] if len(sys.argv) > 1 else "512V"
cyc = sys.argv[2] if len(sys.argv) > 2 else "F"
t0 = time.perf_counter()
grid, model, sfield, _ = bench.build_problem(em, wl, 1.0)
t1 = time.perf_counter()
e, info = em.solve(grid, model, sfield, cycle=cyc, semicoarsening=True, linerelaxation=True, tol=1e-6, verb=0, return_info=True)
t2 = time.perf_counter()
print(f"{wl} {cyc}-cycle: problem built in {t1 - t0:.1f} s (host); solve() {t2 - t1:.2f} s wall, {info['it_mg']} cycles, exit {info['exit']} "
      f"({info['exit_message']}), rel. error {info['rel_error']:.2e}; runtime_at_cycle {np.round(info['runtime_at_cycle'], 2).tolist()}", flush=True)
vm = em.VolumeModel(grid, model, sfield)
from emg3d_amd import _lib
print("device memory:", _lib.mem_info(0))
