#!/bin/bash
# round 6, call B: full GPU suite on the split build; then the placement table (5 fresh processes with the search, 2 without)
mkdir -p gpurun_out/r06/b
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
for i in 1 2 3 4 5; do
  EMG3D_LOG_SETUP=1 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 6 > gpurun_out/r06/b/on_$i.json 2> gpurun_out/r06/b/on_$i.err
done
for i in 1 2; do
  EMG3D_PLACE_TRIES=0 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 6 > gpurun_out/r06/b/off_$i.json 2> gpurun_out/r06/b/off_$i.err
done
grep -h "\[place\]" gpurun_out/r06/b/on_*.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/b/*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(d["per_rank_device"][0]["uuid"][-8:], f.split("/")[-1], "cycle %.2f ms" % d["ms_per_step"], "launch dense %.4f sparse %.4f" % (r["launch_ms"], r["launch_ms_sparse_source"]),
          "sweep_ms", {k: round(v, 3) for k, v in r["sweep_ms"].items()}, "frac %.4f" % r["frac"], "setup %.2f s" % d["setup_plus_warmup_s"],
          "vs_profile", r.get("vs_profile"), "placement", {k: (v.get("tries"), v.get("kept")) for k, v in (r.get("placement") or {}).get("per_working_copy", {}).items()})
PY
