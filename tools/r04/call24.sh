#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for nh in 3 2; do echo "== THA=$nh"; EMG3D_THA=$nh EMG3D_Q_TILE=256 timeout 100 python3 tools/r04/tha_ts.py; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c24_tha_ts.txt
