"""Where the one-off set-up time of a device handle goes (128^3 bench problem): creation, the three
prepare() calls of an sc+lr run (EMG3D_LOG_SETUP=1 prints the split inside the library), close()."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EMG3D_LOG_SETUP", "1")
import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd import models
from emg3d_amd.solver import DeviceMG, MGParameters
grid, model, sfield, cycle = bench.build_problem(em, sys.argv[1] if len(sys.argv) > 1 else "128F", 1.0)
for rep in range(2):
    t = [time.perf_counter()]
    parts = models.eta_factored(grid, model, sfield); t.append(time.perf_counter())
    dev = DeviceMG.from_sigma_volume(grid, *parts[:4], smu0=parts[4]); t.append(time.perf_counter())
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None); t.append(time.perf_counter())
    for sc, lr in ((1, 4), (2, 5), (3, 6)):
        dev.prepare(sc, lr); t.append(time.perf_counter())
    dev.cycle(1, 4); t.append(time.perf_counter())
    dev.close(); t.append(time.perf_counter())
    names = ["eta_factored", "create", "params+source", "prepare(1,4)", "prepare(2,5)", "prepare(3,6)", "one cycle", "close"]
    print(f"rep {rep}: " + ", ".join(f"{n} {1e3 * (b - a):.1f} ms" for n, a, b in zip(names, t[:-1], t[1:])))
