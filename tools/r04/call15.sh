#!/bin/bash
# round 4, call 15: k_line_sweep_thm<RS> parity hunt: plain, produce-then-consume (16), helpers stay for the barriers (32)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for tile in 0 16 32; do
  echo "== EMG3D_Q_TILE=$tile"; EMG3D_Q_TILE=$tile timeout 250 python3 tools/r04/dbg_thr.py 64x48x44 32x66x100 24x130x34
done
echo "== LPW=4"; EMG3D_THR_LPW=4 timeout 250 python3 tools/r04/dbg_thr.py 64x64x64
echo "== LPW=4 tile 16"; EMG3D_THR_LPW=4 EMG3D_Q_TILE=16 timeout 250 python3 tools/r04/dbg_thr.py 64x64x64
} 2>&1 | grep -v amdgpu.ids | tee $O/c15_dbg.txt
