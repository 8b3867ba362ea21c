"""Only batched cycles (for rocprofv3 --kernel-trace --stats): python tools/batch_cycle.py 128F <nsys> [cycles]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

wl, nsys = sys.argv[1], int(sys.argv[2])
ncyc = int(sys.argv[3]) if len(sys.argv) > 3 else 6
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
dev = DeviceMG(grid, vm, sfield.dtype)
dev.set_params(var)
if nsys > 1:
    dev.set_batch(nsys)
rng = np.random.default_rng(1)
for b in range(nsys):
    dev.select(b)
    dev.set_source([rng.uniform(-800, 800), rng.uniform(-800, 800), rng.uniform(-300, 300), rng.uniform(0, 360),
                    rng.uniform(-30, 30)], sfield.smu0)
for sc, lr in zip([1, 2, 3], [4, 5, 6]):
    dev.prepare(sc, lr)
dev.cycles(3, [1, 2, 3], [4, 5, 6]); dev._lib.emg3d_mg_sync(dev._h)
t0 = time.perf_counter()
dev.cycles(ncyc, [1, 2, 3], [4, 5, 6]); dev._lib.emg3d_mg_sync(dev._h)
t = (time.perf_counter() - t0) / ncyc
print(f"{wl} nsys {nsys}: {t*1e3:.2f} ms per cycle, {t*1e3/nsys:.2f} per system, {nsys*grid.nC/t/1e6:.0f} Mcells/s")
dev.close()
