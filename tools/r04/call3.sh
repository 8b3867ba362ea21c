#!/bin/bash
# round 4, call 3: DPP micro; where k_line_sweep_pc spends its time (phases switched off through the lab knob EMG3D_Q_TILE)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 120 tools/micro/dpp_f64 2>&1 | tee $O/c3_dpp_micro.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32"; do
  for nl in 1 2; do for dbg in 0 1 2 3 7; do echo "== $shp: pc NL=$nl dbg=$dbg"; EMG3D_PC_NL=$nl EMG3D_Q_TILE=$dbg timeout 300 python3 tools/sweep_dirs.py $shp; done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c3_phases.txt
