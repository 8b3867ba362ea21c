"""A few isolated level-0 sweeps of a batched handle and nothing else (for rocprofv3 --pmc):
python tools/batch_sweep.py 128F <nsys> [direction] [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

wl, nsys = sys.argv[1], int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
with DeviceMG(grid, vm, sfield.dtype) as dev:
    dev.set_params(var)
    if nsys > 1:
        dev.set_batch(nsys)
    rng = np.random.default_rng(1)
    for b in range(nsys):
        dev.select(b)
        dev.set_source([rng.uniform(-800, 800), rng.uniform(-800, 800), rng.uniform(-300, 300), rng.uniform(0, 360),
                        rng.uniform(-30, 30)], sfield.smu0)
    ms = dev.time_sweep(d, reps)
    print(f"{wl} nsys {nsys} direction {d}: {ms / 4 * 1e3:.1f} us per launch, {ms / 4 / nsys * 1e3:.1f} per system, {dev.last_sweep_kernel()}")
