"""A/B of level-0 sweep variants in one process: python tools/ab_q.py 128F "EMG3D_Q_LPW=4" "EMG3D_Q=0" ..."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

wl = sys.argv[1]
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
keys = set()
for spec in sys.argv[2:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    for k in keys:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = v
        keys.add(k)
    dev = DeviceMG(grid, vm, sfield.dtype)
    dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None)
    r = bench.roofline_of(dev, grid, wl)
    cyc = None
    if "--cycle" in os.environ.get("AB_OPTS", ""):
        for sc, lr in zip([1, 2, 3], [4, 5, 6]):
            dev.prepare(sc, lr)
        dev.cycles(3, [1, 2, 3], [4, 5, 6]); dev._lib.emg3d_mg_sync(dev._h)
        import time
        t0 = time.perf_counter(); dev.cycles(6, [1, 2, 3], [4, 5, 6]); dev._lib.emg3d_mg_sync(dev._h)
        cyc = (time.perf_counter() - t0) / 6 * 1e3
    print(f"{spec or 'default':45s} {r['kernel']:28s} launch {r['launch_ms']*1e3:8.1f} us  frac {r['frac']*100:5.1f}%  sweeps {r['sweep_ms']}  cycle {cyc}", flush=True)
    dev.close()
