#!/bin/bash
# round 6, call D: chain form of the scan kernel on lines of <= 4 / <= 8 blocks (lab knob EMG3D_QPL_CHAIN) against the scans:
# 128^3 F-cycle, alternating, three repetitions; residual histories side by side; kernel averages from rocprofv3
mkdir -p gpurun_out/r06/d
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for rep in 1 2 3; do for c in 0 4 8 16; do
  EMG3D_QPL_CHAIN=$c python bench.py --no-cpu --no-tol --no-256 --batch 0 --no-roofline --steps 30 > gpurun_out/r06/d/chain${c}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json, glob
base = None
for c in (0, 4, 8, 16):
    ms = []; hist = None
    for f in sorted(glob.glob(f"gpurun_out/r06/d/chain{c}_*.json")):
        d = json.load(open(f)); ms.append(round(d["ms_per_step"], 4)); hist = d["rel_error_after"]
    if base is None: base = hist
    dev = max(abs(a - b) / abs(b) for a, b in zip(hist, base))
    print(f"EMG3D_QPL_CHAIN={c}: ms per 128^3 F-cycle {ms}; max rel. deviation of the per-cycle norms from the scan form {dev:.2e}; cycles to 1e-6 {d['cycles_to_1e-6']}")
PY
cd /tmp && export TMPDIR=/tmp
for c in 0 8; do
  EMG3D_QPL_CHAIN=$c rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/d/prof$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-tol --no-256 --batch 0 --no-roofline --steps 30 > /dev/null 2>&1
  echo "== EMG3D_QPL_CHAIN=$c"; f=$(find $GRAFT_REPO_ROOT/gpurun_out/r06/d/prof$c -name "*kernel_stats.csv" | head -1); grep "k_line_sweep_qpl" $f | awk -F, '{print $1, $2, $4}' | cut -c1-220
done
