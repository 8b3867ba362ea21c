#!/bin/bash
# round 5, call 45: the two-sided kernel between 128^3 and 180^3 with two prefetch stages (204 registers: two waves per SIMD fit;
# the three-stage ZS instantiation is 260 = one)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
echo "EMG3D_TW_STAGES=2"; EMG3D_TW_STAGES=2 timeout 900 python3 tools/r05/size_scan.py 128 136 144 152 160 168 176 180 2>/dev/null
echo "EMG3D_TW_STAGES=2 EMG3D_TH_LPW=12"; EMG3D_TW_STAGES=2 EMG3D_TH_LPW=12 timeout 900 python3 tools/r05/size_scan.py 128 136 152 160 176 2>/dev/null
echo "EMG3D_ZSEP=0 (three stages, 252 registers)"; EMG3D_ZSEP=0 timeout 900 python3 tools/r05/size_scan.py 128 136 160 176 2>/dev/null
} | tee $O/c45_thm_stages.txt
