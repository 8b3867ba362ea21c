#!/bin/bash
# round 5, call 38: the full-size GPU tests with the oracle's colour sweeps on host threads: durations
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_batch.py -q -x --durations=12 2>&1 | tail -22 | tee $O/c38_durations.txt
