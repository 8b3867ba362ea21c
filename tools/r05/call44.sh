#!/bin/bash
# round 5, call 44: the two-sided kernel between 128^3 and 180^3: 8 against 12 lines per pair of waves (lab knob EMG3D_TH_LPW), the quad
# kernel there (EMG3D_Q=2)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
echo "EMG3D_TH_LPW=12"; EMG3D_TH_LPW=12 timeout 900 python3 tools/r05/size_scan.py 128 136 144 152 160 168 176 180 2>/dev/null
echo "EMG3D_Q=2 (quad kernel, balanced)"; EMG3D_Q=2 timeout 900 python3 tools/r05/size_scan.py 128 136 152 168 180 2>/dev/null
echo "EMG3D_TH_LPW=4"; EMG3D_TH_LPW=4 timeout 900 python3 tools/r05/size_scan.py 96 112 128 2>/dev/null
} | tee $O/c44_thm_lpw.txt
