"""Config 4 (BASELINE.json configs[3]): 128^3 stretched grid, BiCGSTAB preconditioned by F-cycles with
semicoarsening + line relaxation; device-resident iteration vs SciPy's host iteration."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa (one HIP runtime for both libraries)
import bench
import emg3d_amd as em
from emg3d_amd import solver

wl = sys.argv[1] if len(sys.argv) > 1 else "128F"
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
out = {}
for name, flag in (("device", True), ("host_scipy", False)):
    solver.DEVICE_KRYLOV = flag
    t0 = time.perf_counter()
    e, info = em.solve(grid, model, sfield, return_info=True, sslsolver='bicgstab', cycle=cycle,
                       semicoarsening=True, linerelaxation=True, ordering='colour', verb=0)
    dt = time.perf_counter() - t0
    out[name] = {"s": dt, "it_ssl": info['it_ssl'], "it_mg": info['it_mg'], "exit": info['exit'],
                 "rel_error": info['rel_error']}
    out[name + "_field_norm"] = float(np.linalg.norm(e))
print(json.dumps(out))
