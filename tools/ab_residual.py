"""k_residual A/B: python tools/ab_residual.py [128F|256V ...]  -- times emg3d_mg_time_residual (HIP events, level 0, mode 1)
with the plain and the XCD-aware block map (EMG3D_RES_XCD=0|1, read when the handle is created), 1 / 2 / 4 / 8 node planes
per thread (EMG3D_RES_KZ) and, at 128^3, for a batched handle of 4 systems.  only=<xcd>/<kz>: one setting (PMC runs)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import emg3d_amd as em
import bench
from emg3d_amd.solver import DeviceMG, MGParameters

only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]          # only=0 / only=1: one map (for rocprofv3 --pmc runs)
wls = [a for a in sys.argv[1:] if not a.startswith("only=")]
for wl in (wls or ["128F", "256V"]):
    grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC)
    for nsys in ((1, 4) if wl == "128F" else (1,)):
        for xcd, kz in ([tuple(o.split("/")) for o in only] or
                        (("0", "1"), ("1", "1"), ("2", "1"), ("0", "2"), ("0", "4"), ("0", "8"), ("0", "16"), ("2", "2"), ("2", "4"),
                         ("2", "8"), ("2", "16"), ("1", "8"), ("0", "1"), ("0", "4"))):
            os.environ["EMG3D_RES_XCD"] = xcd
            os.environ["EMG3D_RES_KZ"] = kz
            dev = DeviceMG(grid, vm, sfield.dtype)
            dev.set_params(var)
            if nsys > 1:
                dev.set_batch(nsys)
            for b in range(nsys):
                dev.select(b)
                dev.set_sfield(sfield)
            ms = min(dev.time_residual(10) for _ in range(3))
            print(f"{wl} nsys {nsys} EMG3D_RES_XCD={xcd} EMG3D_RES_KZ={kz}: {ms*1e3:8.1f} us per launch, "
                  f"{200 * grid.nC * nsys / (ms * 1e-3) / 1e9:7.0f} GB/s algorithmic", flush=True)
            dev.close()
