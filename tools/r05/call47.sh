#!/bin/bash
# round 5, call 47: parity tests at 144^3 / 200^3 (launch shapes by rounds of waves)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "between_powers" --durations=3 2>&1 | tail -8 | tee $O/c47_tests.txt
