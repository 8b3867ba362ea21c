"""Lab experiment: forward and backward visiting orders of the line colours chosen independently
(EMG3D_COLOUR_ORDER, EMG3D_COLOUR_ORDER_B).  python tools/colour_order2.py"""
import itertools, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
from emg3d_amd import _lib
import bench

_lib.use(_lib.LAB_PATH)
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
orders = ["".join(p) for p in itertools.permutations("0123")]
res = []
for f in orders:
    for b in orders:
        os.environ["EMG3D_COLOUR_ORDER"] = f
        os.environ["EMG3D_COLOUR_ORDER_B"] = b
        t0 = time.perf_counter()
        e, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0,
                           return_info=True, tol=1e-6, maxit=40)
        t = time.perf_counter() - t0
        err = np.array(info['error_at_cycle']) / info['ref_error']
        rate = (err[-1] / err[1]) ** (1.0 / max(len(err) - 2, 1))
        res.append((rate, info['it_mg'], f, b, t, b[3] == f[0]))
res.sort()
for rate, it, f, b, t, skip in res[:40]:
    print(f"forward {f} backward {b}: {it:2d} cycles, reduction {rate:.3f}, {t:.3f} s, turn-around colour repeated (skipped): {skip}")
print("...")
for rate, it, f, b, t, skip in res[-5:]:
    print(f"forward {f} backward {b}: {it:2d} cycles, reduction {rate:.3f}, {t:.3f} s, skipped: {skip}")
best_skip = [r for r in res if r[5]][:10]
print("best with a skipped turn-around (7 passes per two sweeps):")
for rate, it, f, b, t, skip in best_skip:
    print(f"forward {f} backward {b}: {it:2d} cycles, reduction {rate:.3f}, {t:.3f} s")
