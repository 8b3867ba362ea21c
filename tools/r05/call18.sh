#!/bin/bash
# round 5, call 18: residual kernel choice on the mid levels (lab knob EMG3D_RES_ZM_MIN_CELLS: k_residual_zm from this many cells on; default 2^20)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do for v in "X=1" "EMG3D_RES_ZM_MIN_CELLS=400000" "EMG3D_RES_ZM_MIN_CELLS=100000" "EMG3D_RES_ZM_MIN_CELLS=400000 EMG3D_RES_KZ=2"; do
  echo "$v 128F: $(env $v timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")"
done; done
} 2>&1 | tee $O/c18_res_zm.txt
