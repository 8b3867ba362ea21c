"""k_hfield on the device-resident field of a handle (run under `rocprofv3 --kernel-trace --stats`):
algorithmic bytes = 96 B/cell (48 read + 48 written) + 8 B/cell zeta with mu_r."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
import emg3d_amd as em
from emg3d_amd import models
from emg3d_amd.solver import DeviceMG
name = sys.argv[1] if len(sys.argv) > 1 else "128F"
grid, model, sfield, cycle = bench.build_problem(em, name, 1.0)
parts = models.eta_factored(grid, model, sfield)
with DeviceMG.from_sigma_volume(grid, *parts[:4], smu0=parts[4]) as dev:
    rng = np.random.default_rng(0)
    dev.set_efield(em.Field(grid, rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE), freq=1.0))
    for mu in (False, True):
        for k in range(6):
            t0 = time.perf_counter()
            h = dev.get_hfield(grid, sfield.smu0, mu_r=mu)
            dt = time.perf_counter() - t0
        print(f"{name} mu_r={mu}: get_hfield incl. download {dt * 1e3:.2f} ms ({h.nbytes / dt / 1e9:.1f} GB/s over PCIe), |H|max {np.abs(h).max():.3e}")
