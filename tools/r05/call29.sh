#!/bin/bash
# round 5, call 29: placement of the level-0 arrays against the launch time (lab knob EMG3D_ALLOC_SKEW; tools/r05/bimodal.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
EMG3D_LOG_ALLOC=1 timeout 600 python3 tools/r05/bimodal.py 256V 0 2>&1 | grep -v amdgpu.ids | head -60
for p in 1 2; do echo "process $p"; timeout 900 python3 tools/r05/bimodal.py 256V 0 256 1024 4096 4352 69888 1118464 2097152 0 2>/dev/null; done
echo "128F"; timeout 900 python3 tools/r05/bimodal.py 128F 0 256 1024 4096 4352 69888 1118464 0 2>/dev/null
} | tee $O/c29_skew.txt
