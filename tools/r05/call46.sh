#!/bin/bash
# round 5, call 46: round-aware lines per pair of the two-sided kernel (product library): level-0 sweeps 128^3 ... 180^3; V-cycles at
# 384^3 / 448^3 (their coarse levels take the new rules too); the kernel / variant / batch / full-size tests
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["config"]["cells"], "cells: cycle", round(d["ms_per_step"],2), "ms =", round(d["value"],1), "Mcells/s; launch", round(r["launch_ms"],4), round(r["frac"],4), d["rel_error_after"][-1])'
{
timeout 900 python3 tools/r05/size_scan.py 128 136 144 152 160 176 2>/dev/null
for w in 384V 448V 128F 256V; do echo "$w: $(timeout 600 python3 bench.py --workload $w --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 2>/dev/null | python3 -c "$P")"; done
} | tee $O/c46_thm_rule.txt
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py tests/test_gpu_batch.py tests/test_gpu_solver.py -q -x 2>&1 | tail -3 | tee $O/c46_tests.txt
