#!/bin/bash
# round 3, GPU call 24: what bounds a chain kernel on the level-1 shape of the 128^3 F-cycle (64 x 128 x 64)?
# prefetch depth (2 / 3 stages), lines per pair of waves (4 / 8 / 12), against the default scan kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c24; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
echo "default (scan kernel)"; python3 tools/sweep_dirs.py 64 128 64
for st in 3 2; do for lpw in 4 8 12; do
  echo "chain kernel thm, stages $st, lines per pair $lpw"
  EMG3D_QPL=0 EMG3D_TW_STAGES=$st EMG3D_TH_LPW=$lpw python3 tools/sweep_dirs.py 64 128 64
done; done
echo "quad kernel qc (one-sided), 16 / 8 / 4 / 2 lines per wave"
for lpw in 16 8 4 2; do EMG3D_QPL=0 EMG3D_Q=2 EMG3D_Q_LPW=$lpw python3 tools/sweep_dirs.py 64 128 64; done
echo "32 x 128 x 32 (level 2): default, thm 3 stages 4 lines"
python3 tools/sweep_dirs.py 32 128 32
EMG3D_QPL=0 EMG3D_TH_LPW=4 python3 tools/sweep_dirs.py 32 128 32
EMG3D_QPL=0 EMG3D_TH_LPW=2 python3 tools/sweep_dirs.py 32 128 32
} 2>&1 | grep -v amdgpu.ids | tee $O/mid_level.txt
