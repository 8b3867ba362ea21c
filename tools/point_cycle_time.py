"""Cycle time with the POINT smoother (semicoarsening=False, linerelaxation=False: the defaults of emg3d.solve):
python tools/point_cycle_time.py [128F] [cycles]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

wl = sys.argv[1] if len(sys.argv) > 1 else "128F"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
for sc, lr in ((False, False), (True, False), (False, True)):
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=lr, semicoarsening=sc, vnC=grid.vnC)
    with DeviceMG(grid, vm, sfield.dtype) as dev:
        dev.set_params(var)
        dev.set_sfield(sfield)
        dev.set_efield(None)
        scs = [var.sc_dir] if not var.sc_cycle else [1, 2, 3]
        lrs = [var.lr_dir] if not var.lr_cycle else [4, 5, 6]
        dev.cycles(3, scs, lrs)
        t0 = time.perf_counter()
        norms = dev.cycles(n, scs, lrs)
        dt = (time.perf_counter() - t0) / n
    print(f"{wl} F-cycle sc={sc} lr={lr}: {1e3 * dt:.2f} ms per cycle = {grid.nC / dt / 1e6:.0f} Mcells/s; norms {np.asarray(norms).ravel()[-2:]}", flush=True)
