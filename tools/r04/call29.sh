#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -m gpu -x -k "affine or (fullsize and 128F)" 2>&1 | tail -4
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
EMG3D_THA=3 EMG3D_Q_TILE=256 timeout 100 python3 tools/r04/tha_ts.py | head -6
} 2>&1 | grep -v amdgpu.ids | tee $O/c29_tha_ts.txt
