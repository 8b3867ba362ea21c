#!/bin/bash
# round 5, call 48: level 2 of the 256^3 V-cycle (256 x 64 x 64: 4096 lines of 64 blocks per colour = TWO rounds of k_line_sweep_tha
# workgroups at one per CU): the affine kernel against the two-sided kernel (one round of 1024 waves) and the scan kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
run() { echo "$1: y $(env $2 SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py $3 2 20 2>/dev/null | tail -1)   z $(env $2 SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py $3 3 20 2>/dev/null | tail -1)"; }
{
for g in "256 64 64" "192 64 64" "256 48 48"; do echo "grid $g"
run "default (tha)" "X=1" "$g"
run "thm 8 lines/pair" "EMG3D_THA=0 EMG3D_QPL=0" "$g"
run "thm 12 lines/pair" "EMG3D_THA=0 EMG3D_QPL=0 EMG3D_TH_LPW=12" "$g"
run "scan kernel" "EMG3D_THA=0" "$g"
done
} 2>&1 | tee $O/c48_level2_256V.txt
