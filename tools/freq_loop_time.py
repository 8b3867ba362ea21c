"""Frequency loop on one GPU: shard.solve_frequencies (one handle, re-targeted per frequency: emg3d_mg_set_smu0) against one
fresh handle per frequency.  python tools/freq_loop_time.py [64F 128F]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
from emg3d_amd import shard, models, solver, fields

for wl in (sys.argv[1:] or ["32F", "64F", "128F"]):
    grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
    src = [0., 0., 0., 30., 10.]
    opts = dict(cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
    freqs = bench.FREQS
    parts = models.model_parts(grid, model)

    def fresh():
        out = []
        for f in freqs:
            spec = fields.FrequencySpec(f)
            with solver.DeviceMG.from_model_parts(grid, *parts, smu0=spec.smu0) as dev:
                out.append(solver.solve(grid, None, spec, handle=dev, return_info=True, source=(src, 0), **opts))
        return out
    for rep in range(2):
        t0 = time.perf_counter(); a = fresh(); t1 = time.perf_counter()
        b = shard.solve_frequencies(grid, model, src, freqs, **opts); t2 = time.perf_counter()
    same = all(np.array_equal(np.asarray(x[0]), np.asarray(y[0])) for x, y in zip(a, b))
    its = sum(i['it_mg'] for _, i in b)
    print(f"{wl}: {len(freqs)} frequencies, {its} cycles: fresh handles {1e3 * (t1 - t0):.1f} ms, one handle {1e3 * (t2 - t1):.1f} ms "
          f"({1e3 * (t1 - t0 - (t2 - t1)) / len(freqs):.1f} ms per frequency saved), bitwise equal: {same}", flush=True)
