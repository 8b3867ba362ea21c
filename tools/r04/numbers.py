"""Print the numbers of a profiles/ collection that README / DESIGN / profiles/README quote: python tools/r04/numbers.py [tag]"""
import json, csv, sys, os
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles")
L = lambda f: json.load(open(os.path.join(P, f"{tag}_{f}.json")))
d = L("bench_128F"); r = d["roofline"]
print("128F value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "lex", d["reference_order_lex"])
print(" roofline dense frac", round(r["frac"], 4), "sparse", round(r["frac_sparse_source"], 4), "floor", round(r["formulation_floor_frac"], 4),
      "launch_ms", round(r["launch_ms"], 4), round(r["launch_ms_sparse_source"], 4), "achieved", round(r["achieved"]),
      "rocprof", r["rocprof_average"]["average_ms"], r["rocprof_average_sparse_source"]["average_ms"], "traffic", r["traffic"], r["traffic_sparse_source"], r["traffic_stale"])
print(" cycle_alg", d["cycle_algorithmic"]["frac"], d["cycle_algorithmic"]["frac_executed"], "ttt", d["time_to_tol"])
c = d["config_256V"]; print(" config_256V", c["ms_per_cycle"], c["Mcells_per_s"], c["cycle_algorithmic"]["frac"], c["cycle_algorithmic"]["frac_executed"])
print(" hbm", d["hbm_stream"], "res", d["residual_kernel"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"], "code", d["code"])
v = L("bench_256V"); r = v["roofline"]
print("256V value", round(v["value"], 1), "ms", round(v["ms_per_step"], 3), "frac", round(r["frac"], 4), round(r["frac_sparse_source"], 4), round(r["formulation_floor_frac"], 4),
      "launch", round(r["launch_ms"], 4), round(r["launch_ms_sparse_source"], 4), "rocprof", r["rocprof_average"]["average_ms"], r["rocprof_average_sparse_source"]["average_ms"],
      "traffic", r["traffic"], r["traffic_sparse_source"], "res", v["residual_kernel"], "cyc", v["cycle_algorithmic"]["frac"], v["cycle_algorithmic"]["frac_executed"])
w = L("bench_384V"); r = w["roofline"]
print("384V", round(w["value"], 1), round(w["ms_per_step"], 2), round(r["launch_ms"], 3), round(r["frac"], 4), w["cycle_algorithmic"]["frac"])
x = L("bench_128F_lex"); print("lex", x["ms_per_step"], x["value"])
for e in L("bench_128F_batch")["batched_sources"]:
    print(" batch", e["systems"], e["batch_tune"], round(e["ms_per_cycle_per_system"], 3), round(e["level0_sweep_launch_ms"] * 1e3 / e["systems"], 1), "us/system")
m = L("bench_128F_multi3"); print("multi3", [m[k] for k in m if "concurrent" in k])
t = json.load(open(os.path.join(P, "traffic.json"))); print("traffic", t["source"], {k: (round(t[k]["ratio_to_algorithmic"], 3), round(t[k]["ratio_to_algorithmic_dense_source"], 3)) for k in ("128F", "256V")})
print("sq", json.load(open(os.path.join(P, f"{tag}_sweep_128F_sq_counters.json"))))
for f in (f"{tag}_cycle_128F_kernel_stats.csv", f"{tag}_cycle_256V_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv", f"{tag}_sweep_128F_dense_kernel_stats.csv", f"{tag}_sweep_256V_dense_kernel_stats.csv"):
    rows = list(csv.DictReader(open(os.path.join(P, f)))); tot = sum(float(r["TotalDurationNs"]) for r in rows); print(f)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
