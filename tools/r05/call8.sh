#!/bin/bash
# round 5, call 8: level 1 of the 256^3 V-cycle: quad kernel with 16 lines per wave (512 waves), two-sided kernel with 8 / 12 lines per pair
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
run() { echo "$1: $(env $2 timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense 2>/dev/null | python3 -c "$P")"; }
{
for rep in 1 2 3; do
  run "default            " "X=1"
  run "qc 16 lines on L1  " "EMG3D_Q_LPW=16"
  run "thm L1, 8 per pair " "EMG3D_Q_MIN_LINES=8193 EMG3D_TWIST_MAX=8193"
  run "thm L1, 12 per pair" "EMG3D_Q_MIN_LINES=8193 EMG3D_TWIST_MAX=8193 EMG3D_TH_LPW=12"
  run "thm L1, 2 stages   " "EMG3D_Q_MIN_LINES=8193 EMG3D_TWIST_MAX=8193 EMG3D_TW_STAGES=2"
done
} 2>&1 | tee $O/c8_level1_variants.txt
