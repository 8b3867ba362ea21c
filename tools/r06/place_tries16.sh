#!/bin/bash
# how often does a longer search (16 candidates) find the fast class where 12 do not?  3 fresh processes, set-up only
mkdir -p gpurun_out/r06/p16
for i in 1 2 3; do
  EMG3D_PLACE_TRIES=16 EMG3D_LOG_SETUP=1 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 3 --no-roofline > gpurun_out/r06/p16/on_$i.json 2> gpurun_out/r06/p16/on_$i.err
  grep -h "\[place\]" gpurun_out/r06/p16/on_$i.err
  python - gpurun_out/r06/p16/on_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["per_rank_device"][0]["uuid"][-8:], "cycle %.2f ms" % d["ms_per_step"], "setup %.2f s" % d["setup_plus_warmup_s"])
PY
done
