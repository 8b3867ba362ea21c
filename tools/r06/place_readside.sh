#!/bin/bash
# lab experiment (EMG3D_PLACE_X=1): where no candidate of the field copy stands out, re-roll the blocks the sweep reads (source copy, factor)
mkdir -p gpurun_out/r06/px
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for i in 1 2 3; do
  EMG3D_PLACE_X=1 EMG3D_LOG_SETUP=1 timeout 300 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 3 > gpurun_out/r06/px/on_$i.json 2> gpurun_out/r06/px/on_$i.err
  grep -h "\[place" gpurun_out/r06/px/on_$i.err
  python - gpurun_out/r06/px/on_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(d["per_rank_device"][0]["uuid"][-8:], "cycle %.2f ms" % d["ms_per_step"], "launch dense %.4f" % r["launch_ms"], {k: round(v, 3) for k, v in r["sweep_ms"].items()}, "setup %.2f s" % d["setup_plus_warmup_s"])
PY
done
