"""Level-0 sweep efficiency against the grid size (HISTORY R5.19): is the launch slower where the three components of the field array
lie at distances (mod 128 MiB) that the copy micro-benchmark marks as bad for a write stream against a read stream?
python tools/r05/size_scan.py n [n ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import emg3d_amd as em
from emg3d_amd import _lib
from emg3d_amd.solver import DeviceMG, MGParameters

for n in [int(a) for a in sys.argv[1:]]:
    ncore = n // 2
    npad = (n - ncore) // 2
    ncore = n - 2 * npad
    fac = float(12.5 ** (1.0 / npad))
    bench.WORKLOADS["scan"] = (n, ncore, npad, 25. * 256 / n, (fac, fac, fac * 1.002), 'V')
    grid, model, sfield, cycle = bench.build_problem(em, "scan", 1.0)
    assert grid.vnC[0] == n, grid.vnC
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')
    rng = np.random.default_rng(5)
    dense = (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-9
    comp = n * (n + 1) * (n + 1) * 16 / 2 ** 20
    ds = sorted(round((k * comp) % 128, 1) for k in (-2, -1, 1, 2))
    _lib.load().emg3d_hip_release_cached()
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var)
        dev.set_sfield(dense)
        dev.set_efield(None)
        for d in (1, 2, 3):
            dev.time_sweep(d, 1)
        t = {d: float(np.median([dev.time_sweep(d, 2) for _ in range(4)])) / 4 for d in (1, 2, 3)}       # ms per launch
        alg = 200.0 * n ** 3 / 4
        fr = {d: alg / (t[d] * 1e-3) / 8e12 for d in t}
        print(f"n = {n}: component {comp:7.1f} MiB, distances mod 128 MiB {ds} {'BAD' if any(10 < x < 68 for x in ds) else 'ok '}  "
              f"launch ms x/y/z {t[1]:.3f} {t[2]:.3f} {t[3]:.3f}  frac {fr[1]:.4f} {fr[2]:.4f} {fr[3]:.4f}  mean {np.mean(list(fr.values())):.4f}  "
              f"{dev.last_sweep_kernel()}", flush=True)
    del dense, vm, grid, model, sfield
