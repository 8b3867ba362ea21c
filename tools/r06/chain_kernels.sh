#!/bin/bash
# round 6, call E: per-kernel averages of the 128^3 F-cycle with the chain form of the scan kernel on lines of <= 4 / <= 8 blocks (lab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06/e
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for c in 0 4 8; do
  EMG3D_QPL_CHAIN=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/e/prof$c -- python3 bench.py --steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-roofline > gpurun_out/r06/e/prof$c.log 2>&1
  echo "== EMG3D_QPL_CHAIN=$c"; f=$(find gpurun_out/r06/e/prof$c -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_line_sweep_qpl" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
  find gpurun_out/r06/e/prof$c -type f ! -name '*kernel_stats.csv' -delete
done
