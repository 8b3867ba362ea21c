#!/bin/bash
# round 3, GPU call 19: source-line flags for batched systems -- parity and the batched figure
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c19; mkdir -p $O
LAB=$PWD/emg3d_amd/libemg3d_hip_lab.so
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_batch.py -q -m gpu -x -k "batch or source" > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for rep in 1 2; do for f in 1 0; do
  EMG3D_SFLAG=$f EMG3D_HIP_LIB=$LAB timeout 600 python3 tools/batch_cycle.py 128F 8 9 2>&1 | tail -1 | sed "s/^/sflag=$f /"
done; done | tee $O/ab_batch.txt
