"""(The byte-count arguments need the throwaway EMG3D_ALLOC_SKEW patch of HISTORY R5.18, which was not kept.)
Is the 7 % spread of the 256^3 level-0 launch between processes a property of where the arrays were allocated?
One process, the handle rebuilt several times.  Without arguments beyond the workload: (a) blocks re-used from the library's pool (same
addresses), (b) pool released to the driver in between, (c) a dummy allocation of odd size placed in front.  With a list of byte
counts (lab build): the k-th large allocation of a handle is placed k * skew (mod 2 MiB) bytes into its block (EMG3D_ALLOC_SKEW).
Prints the dense-source launch time per direction."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import emg3d_amd as em
from emg3d_amd import _lib
from emg3d_amd.solver import DeviceMG, MGParameters
import torch

wl = sys.argv[1] if len(sys.argv) > 1 else "256V"
skews = [int(a) for a in sys.argv[2:]]
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')
rng = np.random.default_rng(5)
dense = (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-9
lib = _lib.load()
dummies = []
modes = [("skew", s) for s in skews] if skews else [(m, 0) for m in ["first", "pool", "pool", "released", "released", "dummy", "dummy", "dummy"]]
for mode, skew in modes:
    if mode in ("released", "skew"):
        lib.emg3d_hip_release_cached()
    if mode == "skew":
        os.environ["EMG3D_ALLOC_SKEW"] = str(skew)
    if mode == "dummy":
        lib.emg3d_hip_release_cached()
        dummies.append(torch.empty(int(rng.integers(1, 400)) * (1 << 20) + 4096 * int(rng.integers(1, 100)), dtype=torch.uint8, device="cuda"))
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var)
        dev.set_sfield(dense)
        dev.set_efield(None)
        for d in (1, 2, 3):
            dev.time_sweep(d, 1)
        t = {d: np.median([dev.time_sweep(d, 2) for _ in range(5)]) / 4 * 1e3 for d in (1, 2, 3)}
        print(f"{mode:9s} {skew:8d} launch us x/y/z: {t[1]:.1f} {t[2]:.1f} {t[3]:.1f}  mean {np.mean(list(t.values())):.1f}  {dev.last_sweep_kernel()}", flush=True)
