#!/bin/bash
# round 4, call 18: RS against the PLAIN two-sided kernel (EMG3D_QPL=0 EMG3D_THR=0) on the same mid-level shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
export EMG3D_THR_MIN_LINES=256 EMG3D_THR_MIN=24
{
for shp in "128 64 64" "64 64 64" "48 48 96" "40 80 80" "32 128 128"; do
  echo -n "qpl   "; EMG3D_THR=0 timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "thm   "; EMG3D_THR=0 EMG3D_QPL=0 timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "thm/4 "; EMG3D_THR=0 EMG3D_QPL=0 EMG3D_TH_LPW=4 timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "RS/8  "; EMG3D_THR_LPW=8 timeout 200 python3 tools/sweep_dirs.py $shp
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c18_thr_vs_thm.txt
