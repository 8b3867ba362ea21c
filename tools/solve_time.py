"""Wall time of complete solve() calls (handle set-up, source upload, cycles, download) per grid size."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
import bench
for wl in ("32F", "64F", "128F"):
    grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
    kw = dict(cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0, return_info=True)
    em.solve(grid, model, sfield, **kw)              # warm-up (pool, code objects)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e, info = em.solve(grid, model, sfield, **kw); ts.append(time.perf_counter() - t0)
    t_src = []
    for _ in range(3):
        t0 = time.perf_counter()
        e2, info2 = em.solve(grid, model, em.fields.FrequencySpec(1.0), source=([0., 0., 0., 30., 10.], 0), download=False, **kw)
        t_src.append(time.perf_counter() - t0)
    print(f"{wl}: solve {min(ts)*1e3:.1f} ms ({info['it_mg']} cycles, {min(ts)*1e3/info['it_mg']:.2f} ms per cycle incl. set-up); "
          f"source in HBM + no download {min(t_src)*1e3:.1f} ms")
