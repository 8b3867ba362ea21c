#!/bin/bash
# round 5, call 3: where a tiny-level launch of k_line_sweep_qpl spends its cycles (lab build, in-kernel stamps: EMG3D_Q_TILE=512)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shape in "128 4 4" "128 8 8" "128 16 16" "128 32 32"; do for d in 1 2 3; do
  echo "== $shape dir $d"
  SWEEP_ONCE_COARSE=1 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py $shape $d 2 2>&1 | grep -v amdgpu.ids | tail -14
done; done
} > $O/c3_qpl_stamps.txt 2>&1
tail -5 $O/c3_qpl_stamps.txt
