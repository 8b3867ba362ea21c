#!/bin/bash
# Placement of level 0's working copies (HISTORY R6.1): fresh processes of the 256^3 bench with and without the search.
# usage: tools/r06/place_probe.sh <out-dir> <tries> <processes>
out=${1:-gpurun_out/r06/place}; tries=${2:-8}; n=${3:-3}
mkdir -p $out
for i in $(seq 1 $n); do
  EMG3D_PLACE_TRIES=$tries EMG3D_LOG_SETUP=1 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 6 \
      > $out/on_$i.json 2> $out/on_$i.err
  EMG3D_PLACE_TRIES=0 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 6 \
      > $out/off_$i.json 2> $out/off_$i.err
done
grep -h "\[place\]" $out/on_*.err
python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f.split("/")[-1], "cycle %.2f ms" % d["ms_per_step"], "launch dense %.4f sparse %.4f" % (r["launch_ms"], r["launch_ms_sparse_source"]),
          "sweep_ms", {k: round(v, 3) for k, v in r["sweep_ms"].items()}, "frac %.4f" % r["frac"], "setup %.2f s" % d["setup_plus_warmup_s"],
          "norm", d["rel_error_after"][-1])
PY
