#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for nh in 3 2; do echo "== THA=$nh"; EMG3D_THA=$nh EMG3D_Q_TILE=256 timeout 100 python3 tools/r04/tha_ts.py | head -5; done
for nh in 3 2; do echo "== THA=$nh"; EMG3D_THA=$nh timeout 250 python3 tools/r04/dbg_thr.py 64x70x66 34x67x69 72x47x66 70x68x51 | grep -v "e-1[0-9]"; done
for shp in "128 64 64" "40 80 80"; do
  for nh in 0 2 3; do echo -n "THA=$nh "; EMG3D_THA=$nh timeout 200 python3 tools/sweep_dirs.py $shp; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c25_tha.txt
