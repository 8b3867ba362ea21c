#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for nh in 0 3; do echo -n "THA=$nh "; EMG3D_THA=$nh timeout 200 python3 tools/sweep_dirs.py 128 64 64; done
for rep in 1 2; do for nh in 0 3; do
  echo -n "THA=$nh 128F cycle: "; EMG3D_THA=$nh timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c31_tha.txt
