#!/bin/bash
# round 4, call 19: is the RS chain wave waiting for its loads?  dbg: factor rows of the forward steps from one block (64), the
# backward steps' loads from one block (128), both (192)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
export EMG3D_THR_MIN_LINES=256 EMG3D_THR_MIN=24 EMG3D_THR_LPW=8
{
for shp in "128 64 64" "40 80 80"; do
  for t in 0 64 128 192 16; do echo -n "tile=$t "; EMG3D_Q_TILE=$t timeout 200 python3 tools/sweep_dirs.py $shp; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c19_rs_loads.txt
