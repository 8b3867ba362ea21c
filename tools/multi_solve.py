"""Aggregate throughput of K independent solves sharing ONE GPU (each handle has its own stream; one host
thread per handle drives emg3d_mg_cycles, ctypes releases the GIL): the coarse levels of a cycle leave most
SIMDs idle, a second frequency fills them."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd import models
from emg3d_amd.solver import DeviceMG, MGParameters
name = sys.argv[1] if len(sys.argv) > 1 else "128F"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ncyc = 12
grid, model, sfield, cycle = bench.build_problem(em, name, 1.0)
devs = []
for k in range(kmax):
    sf = em.get_source_field(grid, [0., 0., -950., 10. * k, 5.], freq=1.0 * (k + 1))
    parts = models.eta_factored(grid, model, sf)
    dev = DeviceMG.from_sigma_volume(grid, *parts[:4], smu0=parts[4])
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    dev.set_params(var); dev.set_sfield(sf); dev.set_efield(None)
    for sc, lr in ((1, 4), (2, 5), (3, 6)):
        dev.prepare(sc, lr)
    devs.append(dev)
devs[0].time_residual(100)       # warm-up
sc, lr = [1, 2, 3], [4, 5, 6]
for k in range(1, kmax + 1):
    for d in devs[:k]:
        d.set_efield(None)
    out = [None] * k
    def run(i):
        out[i] = devs[i].cycles(ncyc, sc, lr)
    for rep in range(2):
        th = [threading.Thread(target=run, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    print(f"{name}: {k} concurrent solves: {dt / ncyc * 1e3:.2f} ms per cycle round, aggregate {k * grid.nC * ncyc / dt / 1e6:.1f} Mcells/s "
          f"(last norms {[f'{o[-1]:.2e}' for o in out]})")
for d in devs: d.close()
