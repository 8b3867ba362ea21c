#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "chain_kernels_colour" 2>&1 | tail -15 | tee $O/c9_tests.txt
