#!/bin/bash
# round 5, call 20: what-if for two systems per wave on batched level-0 sweeps: every second system skips the factor-row loads
# (-DEMG3D_WHATIF_SHAREW: wrong results; the L1 request count of a kernel that shares the rows of a step between two systems in registers)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
{
for rep in 1 2; do for lib in lab sharew now; do for n in 1 2 8; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_$lib.so
  echo "$lib: $(timeout 300 python3 tools/batch_sweep.py 128F $n 3 5 2>/dev/null | tail -1)"
done; done; done
} 2>&1 | tee $O/c20_whatif_sharew.txt
