#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver.py -q -m gpu -x -k "colour" 2>&1 | tail -12 | tee $O/c23_tests.txt
