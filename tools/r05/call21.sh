#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_logging.py -q -m gpu -x 2>&1 | tail -3 | tee $O/c21_tests.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
for rep in 1 2; do echo "128F: $(timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")"; echo "256V: $(timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")"; done | tee -a $O/c21_tests.txt
