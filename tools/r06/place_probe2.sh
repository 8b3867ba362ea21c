#!/bin/bash
# usage: place_probe2.sh <out-dir> <tries> <processes> <what-list>   (what: EMG3D_PLACE_WHAT bit masks, e.g. "1 5 7")
out=${1:-gpurun_out/r06/place2}; tries=${2:-6}; n=${3:-2}; whats=${4:-"1 5 7"}
mkdir -p $out
for i in $(seq 1 $n); do
  for w in $whats; do
    t=$tries; if [ $w = 0 ]; then t=0; fi
    EMG3D_PLACE_WHAT=$w EMG3D_PLACE_TRIES=$t EMG3D_LOG_SETUP=1 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 6 \
        > $out/what${w}_$i.json 2> $out/what${w}_$i.err
  done
done
for f in $out/*.err; do echo "== $f"; grep -h "\[place\]" $f; done
python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(d["per_rank_device"][0]["host"], d["per_rank_device"][0]["uuid"][-8:], f.split("/")[-1], "cycle %.2f ms" % d["ms_per_step"], "launch dense %.4f sparse %.4f" % (r["launch_ms"], r["launch_ms_sparse_source"]),
          "sweep_ms", {k: round(v, 3) for k, v in r["sweep_ms"].items()}, "frac %.4f" % r["frac"], "setup %.2f s" % d["setup_plus_warmup_s"],
          "norm", d["rel_error_after"][-1])
PY
