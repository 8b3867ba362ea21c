"""128^3 F-cycle: per-cycle residual norms and field of the device (both orderings, lane-group vs quad kernels) against
the oracle's SAME ordering (2 cycles, strict build).  Evidence for DESIGN 4 (parity at BASELINE sizes)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import emg3d_amd as em
from oracle import oracle as orc
import bench

ncyc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wl = sys.argv[2] if len(sys.argv) > 2 else "128F"
grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
vm = em.VolumeModel(grid, model, sfield)
om = orc.Mesh(grid.h, grid.origin); ov = orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)
for ordering, order in (("lex", 0), ("colour", 1)):
    t0 = time.perf_counter()
    oe, oinfo = orc.solve(om, ov, np.array(sfield), cycle=cycle, semicoarsening=True, linerelaxation=True, maxit=ncyc, tol=1e-30, order=order)
    print(f"oracle {ordering}: {time.perf_counter()-t0:.1f} s", oinfo['error_at_cycle'], flush=True)
    for env in ({"EMG3D_Q": "0"}, {"EMG3D_Q": "1"}):
        os.environ.update(env)
        e, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, maxit=ncyc, tol=1e-30,
                           return_info=True, verb=0, ordering=ordering)
        dn = np.abs(info['error_at_cycle'] / oinfo['error_at_cycle'] - 1)
        print(f"  {ordering:6s} {env}: per-cycle norm rel.dev {dn}, field relerr {np.abs(np.array(e)-oe).max()/np.abs(oe).max():.2e}", flush=True)
