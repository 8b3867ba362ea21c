#!/bin/bash
# round 4, call 17: where k_line_sweep_thm<RS> wins: isolated sweeps per direction on the shapes of the 128^3 F-cycle's levels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
export EMG3D_THR_MIN_LINES=256 EMG3D_THR_MIN=24
{
for shp in "64 64 32" "64 32 64" "32 64 64" "64 64 64" "48 48 96" "40 80 80" "64 48 48" "56 56 56"; do
  echo -n "THR=0 "; EMG3D_THR=0 timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "LPW=4 "; EMG3D_THR_LPW=4 timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "LPW=8 "; EMG3D_THR_LPW=8 timeout 200 python3 tools/sweep_dirs.py $shp
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c17_thr_shapes.txt
