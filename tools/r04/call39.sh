#!/bin/bash
# round 4, call 39: two blocks per quad in the scan kernel from 32- / 16-block lines on (EMG3D_QPL_M2), cycle A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for rep in 1 2; do for m2 in 64 32 16; do
  echo -n "QPL_M2=$m2 128F cycle: "; EMG3D_QPL_M2=$m2 timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c39_qpl_m2.txt
