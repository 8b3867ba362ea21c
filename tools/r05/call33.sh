#!/bin/bash
# round 5, call 33: the rewritten host control (max_level, _solver_and_cycle, _terminate) under the GPU log / solver tests; a complete
# solve() at 512^3 and 448^3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_logging.py tests/test_gpu_solver.py tests/test_gpu_krylov.py -q -x 2>&1 | tail -3 | tee $O/c33_tests.txt
{
timeout 900 python3 tools/r05/solve512.py 512V F 2>/dev/null
timeout 900 python3 tools/r05/solve512.py 512V V 2>/dev/null
timeout 900 python3 tools/r05/solve512.py 256V F 2>/dev/null
} | tee $O/c33_solve512.txt
