"""Three handles of a size at which the placement search runs (200^3: working copies of 385 MB), driven by three host threads at once
(shard.solve_frequencies(concurrent=3)) against one at a time: bit-identical fields, no dead-lock, pool intact."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd import shard, _lib
grid, model, sfield, cycle = bench.build_problem(em, "200V", 1.0)
freqs = [1.0, 0.5, 2.0]
out = {}
for conc in (3, 1):
    _lib.load().emg3d_hip_release_cached()
    t0 = time.perf_counter()
    res = shard.solve_frequencies(grid, model, [0., 0., 0., 30., 10.], freqs, concurrent=conc, cycle=cycle, semicoarsening=True,
                                  linerelaxation=True, maxit=3, verb=0)
    out[conc] = [(np.array(e), info['it_mg'], float(info['abs_error'])) for e, info in res]
    print("concurrent", conc, "%.2f s" % (time.perf_counter() - t0), [(i, "%.3e" % a) for _, i, a in out[conc]], flush=True)
for (e3, i3, a3), (e1, i1, a1) in zip(out[3], out[1]):
    assert i3 == i1 and a3 == a1 and np.array_equal(e3, e1)
print("bit-identical: OK; pool holds %.1f GB" % (_lib.load().emg3d_hip_cached_bytes() / 1e9))
