#!/bin/bash
# round 5, call 31: placement of the level-0 arrays, coarse skews (multiples of 2 MiB up to 256 MiB between consecutive large arrays)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
export EMG3D_ALLOC_SKEW_MOD=268435456
{
for p in 1 2; do echo "process $p"
timeout 900 python3 tools/r05/bimodal.py 256V 0 2097152 6291456 14680064 31457280 65011712 132120576 2162688 35651584 0 2>/dev/null
done
} | tee $O/c31_skew_coarse.txt
