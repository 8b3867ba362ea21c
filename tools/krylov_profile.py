import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch, bench, emg3d_amd as em
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
kw = dict(return_info=True, sslsolver='bicgstab', cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0)
em.solve(grid, model, sfield, **kw)
t0 = time.perf_counter(); e, info = em.solve(grid, model, sfield, **kw); print("solve", time.perf_counter() - t0, info['it_ssl'], info['it_mg'], info['rel_error'])
pr = cProfile.Profile(); pr.enable(); em.solve(grid, model, sfield, **kw); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
