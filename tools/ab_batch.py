"""Level-0 sweep of a batched handle: python tools/ab_batch.py 128F <nsys> [ENV=V,ENV=V ...]
(for rocprofv3 --pmc FETCH_SIZE / --kernel-trace runs: only sweeps are launched)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

wl, nsys = sys.argv[1], int(sys.argv[2])
for spec in sys.argv[3:]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        os.environ[k] = v
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
dev = DeviceMG(grid, vm, sfield.dtype)
dev.set_params(var)
if nsys > 1:
    dev.set_batch(nsys)
for b in range(nsys):
    dev.select(b)
    dev.set_sfield(sfield)
ms = {d: dev.time_sweep(d, 5) for d in (1, 2, 3)}
print(f"nsys {nsys} {' '.join(sys.argv[3:]):30s} {dev.last_sweep_kernel():28s} launch {sum(ms.values())/12*1e3:8.1f} us "
      f"per system {sum(ms.values())/12/nsys*1e3:8.1f} us   sweeps {ms}", flush=True)
dev.close()
