#!/bin/bash
# round 5, call 6: is the scan kernel's load phase on the 32-block level bound by the number of cache lines per load instruction?
# 32-block lines, ~1000 per colour: y-lines on 64 x 32 x 64 (blocks of a line 1 KB apart: what the cycle runs) against x-lines on
# 32 x 64 x 64 WITHOUT the transposed working copy (EMG3D_XT=0: blocks of a line contiguous); in-kernel stamps (EMG3D_Q_TILE=512)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
echo "== y-lines 64 x 32 x 64"; SWEEP_ONCE_COARSE=1 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py 64 32 64 2 2 2>&1 | grep -v amdgpu.ids | tail -9
echo "== x-lines 32 x 64 x 64, transposed copy (default)"; SWEEP_ONCE_COARSE=1 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py 32 64 64 1 2 2>&1 | grep -v amdgpu.ids | tail -9
echo "== x-lines 32 x 64 x 64, EMG3D_XT=0"; SWEEP_ONCE_COARSE=1 EMG3D_XT=0 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py 32 64 64 1 2 2>&1 | grep -v amdgpu.ids | tail -9
echo "== timing without stamps: y / x(XT) / x(XT=0)"
SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 64 32 64 2 20 2>&1 | tail -1
SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 32 64 64 1 20 2>&1 | tail -1
SWEEP_ONCE_COARSE=1 EMG3D_XT=0 timeout 120 python3 tools/sweep_once.py 32 64 64 1 20 2>&1 | tail -1
} > $O/c6_qpl_coalescing.txt 2>&1
cat $O/c6_qpl_coalescing.txt
