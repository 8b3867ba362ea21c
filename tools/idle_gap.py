"""Which idle gap before a burst of cycles triggers the ~70 ms one-off stall seen in solve()?  One persistent
handle, bursts of 9 cycles separated by host sleeps; prints the excess over 9 x the steady cycle time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd import models
from emg3d_amd.solver import DeviceMG, MGParameters
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
parts = models.eta_factored(grid, model, sfield)
dev = DeviceMG.from_sigma_volume(grid, *parts[:4], smu0=parts[4])
dev.set_params(MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC))
dev.set_sfield(sfield); dev.set_efield(None)
sc, lr = [1, 2, 3], [4, 5, 6]
for s_, l_ in zip(sc, lr):
    dev.prepare(s_, l_)
dev.time_residual(100)
base = None
for gap in (0, 0, 2, 5, 10, 20, 50, 100, 300, 0, 20, 20, 0):
    time.sleep(gap / 1e3)
    t0 = time.perf_counter(); dev.cycles(9, sc, lr); dt = (time.perf_counter() - t0) * 1e3
    if base is None: base = dt
    print(f"idle {gap:4d} ms before: 9 cycles {dt:7.1f} ms (excess {dt - base:6.1f})")
import numpy as np
print("--- which host-side step before a burst triggers the stall?")
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
steps = {
    "nothing": lambda: None,
    "set_efield(None)": lambda: dev.set_efield(None),
    "set_sfield (H2D 102 MB)": lambda: dev.set_sfield(sfield),
    "get_efield (D2H 102 MB)": lambda: dev.get_efield(),
    "set_params": lambda: dev.set_params(var),
    "sfield_norm": lambda: dev.sfield_norm() if hasattr(dev, "sfield_norm") else None,
    "host: 100 MB numpy alloc+free": lambda: np.ones(6_400_000, dtype=np.complex128).sum(),
}
for name, fn in steps.items():
    for rep in range(3):
        fn()
        t0 = time.perf_counter(); dev.cycles(9, sc, lr); dt = (time.perf_counter() - t0) * 1e3
        print(f"{name:32s}: 9 cycles {dt:7.1f} ms (excess {dt - base:6.1f})")
def nine():
    ts = []
    for i in range(9):
        t0 = time.perf_counter(); dev.cycle(sc[i % 3], lr[i % 3]); ts.append((time.perf_counter() - t0) * 1e3)
    return " ".join(f"{t:.1f}" for t in ts)
print("--- A: set_params + 9 x cycle()")
for rep in range(3):
    dev.set_params(var); print("A:", nine())
print("--- B: set_sfield + set_efield(None) + residual_norm + 9 x cycle()")
for rep in range(3):
    dev.set_sfield(sfield); dev.set_efield(None); dev.residual_norm(); print("B:", nine())
print("--- C: B + get_efield into a fresh np.zeros")
for rep in range(3):
    dev.set_sfield(sfield); dev.set_efield(None); dev.residual_norm(); r = nine()
    e = np.zeros(grid.nE, dtype=complex); dev.get_efield(e); print("C:", r)
print("--- D: A + C")
for rep in range(3):
    dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None); dev.residual_norm(); r = nine()
    e = np.zeros(grid.nE, dtype=complex); dev.get_efield(e); print("D:", r)
print("--- E: D with the result array allocated BEFORE the cycles (np.zeros, untouched pages), as solve() does")
for rep in range(3):
    dev.set_params(var); dev.set_sfield(sfield)
    e = em.Field(grid, dtype=sfield.dtype, freq=sfield._freq)
    dev.set_efield(None); dev.residual_norm(); r = nine()
    dev.get_efield(np.asarray(e)); print("E:", r)
print("--- F: E + a fresh MGParameters per solve")
for rep in range(3):
    v2 = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    dev.set_params(v2); dev.set_sfield(sfield)
    e = em.Field(grid, dtype=sfield.dtype, freq=sfield._freq)
    dev.set_efield(None); dev.residual_norm(); r = nine()
    dev.get_efield(np.asarray(e)); print("F:", r)
print("--- G: D + np.linalg.norm(sfield) first (BLAS, multi-threaded)")
for rep in range(3):
    nrm = float(np.linalg.norm(sfield))
    dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None); dev.residual_norm(); r = nine()
    e = np.zeros(grid.nE, dtype=complex); dev.get_efield(e); print("G:", r)
print("--- H: G with BLAS limited to one thread")
from threadpoolctl import threadpool_limits, threadpool_info
print([ (d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
with threadpool_limits(limits=1):
    for rep in range(3):
        nrm = float(np.linalg.norm(sfield))
        dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None); dev.residual_norm(); r = nine()
        e = np.zeros(grid.nE, dtype=complex); dev.get_efield(e); print("H:", r)
    for rep in range(3):
        t0 = time.perf_counter()
        em.solve(grid, None, sfield, handle=dev, return_info=True, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
        print(f"solve with one BLAS thread {1e3 * (time.perf_counter() - t0):.1f} ms")
print("--- solve(handle=dev)")
for rep in range(3):
    t0 = time.perf_counter()
    em.solve(grid, None, sfield, handle=dev, return_info=True, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
    print(f"solve {1e3 * (time.perf_counter() - t0):.1f} ms")
