#!/bin/bash
# round 5, call 25: what bounds k_line_sweep_tha on the 64-block level?  what-if builds (wrong results): no factor-row loads (NOW), no
# neighbour-value loads (NOE) in the helper waves; sweep of 128 x 64 x 64 (4 launches), coarse-level conditions
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
{
for rep in 1 2; do for lib in lab wNOW wNOE; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_$lib.so
  echo "$lib: y $(SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 128 64 64 2 20 2>/dev/null | tail -1)   z $(SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 128 64 64 3 20 2>/dev/null | tail -1)"
done; done
} 2>&1 | tee $O/c25_tha_whatif.txt
