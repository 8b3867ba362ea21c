#!/bin/bash
# round 5, call 40: lines dealt evenly over the SIMDs in the quad kernel (MG::q_balanced_lpw; lab knob EMG3D_Q_BALANCE) against 16 lines
# per wave, by grid size; two against three prefetch stages where a launch is 1-2 rounds
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for b in 0 1; do echo "EMG3D_Q_BALANCE=$b"; EMG3D_Q_BALANCE=$b timeout 1500 python3 tools/r05/size_scan.py 256 288 320 352 368 384 416 448 480 512 2>/dev/null; done
for b in 0 1; do echo "EMG3D_Q_BALANCE=$b EMG3D_Q_STAGES=2"; EMG3D_Q_STAGES=2 EMG3D_Q_BALANCE=$b timeout 1500 python3 tools/r05/size_scan.py 256 288 320 352 384 2>/dev/null; done
} | tee $O/c40_balance.txt
