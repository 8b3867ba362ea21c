#!/bin/bash
# round 5, call 15: fewer, fatter workgroups on the tiny-level launches of the scan kernel (lab knob EMG3D_QPL_NW: independent waves per workgroup)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do for v in "EMG3D_QDESC=0 EMG3D_QPL_NW=1" "EMG3D_QDESC=0 EMG3D_QPL_NW=2" "EMG3D_QDESC=0 EMG3D_QPL_NW=4" "EMG3D_QDESC=0 EMG3D_QPL_NW=8" "EMG3D_QPL_NW=1"; do
  echo "$v 128F: $(env $v timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")"
done; done
} 2>&1 | tee $O/c15_qpl_nw.txt
