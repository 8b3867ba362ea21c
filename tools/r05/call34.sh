#!/bin/bash
# round 5, call 34: ONE slab for the large arrays of a handle + controlled offsets between them (experiment library, patch of
# MG::raw_alloc described in HISTORY R5.18; not part of the product): is there a placement that is reproducibly fast?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/tools/r05/exp/libemg3d_hip_exp.so
export EMG3D_SLAB_GB=44 EMG3D_ALLOC_SKEW_MOD=268435456
{
for p in 1 2; do echo "process $p"
timeout 900 python3 tools/r05/bimodal.py 256V 0 2097152 4194304 6291456 10485760 18874368 35651584 69206016 136314880 2129920 2101248 0 2>/dev/null
done
} | tee $O/c34_slab_skew.txt
