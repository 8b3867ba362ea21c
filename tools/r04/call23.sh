#!/bin/bash
# round 4, call 23: k_line_sweep_tha with the chain waves alone on their SIMDs (SP)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
echo "== THA=3 SP"; EMG3D_THA_SP=1 EMG3D_THA=3 timeout 250 python3 tools/r04/dbg_thr.py 64x70x66 34x67x69 72x47x66 | grep -v "e-1[0-9]"
for shp in "128 64 64" "40 80 80"; do
  for sp in 0 1; do for nh in 2 3 4; do echo -n "SP=$sp THA=$nh "; EMG3D_THA_SP=$sp EMG3D_THA=$nh timeout 200 python3 tools/sweep_dirs.py $shp; done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c23_tha_sp.txt
