#!/bin/bash
# round 5, call 26: fields beyond 4 GiB -- k_line_sweep_qc<..., BIG> (64-bit field offsets): lab parity at small sizes, the 448^3
# product test, 448^3 / 512^3 V-cycle timings (2 vs 3 prefetch stages through the lab build)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_variants.py -q -x -k "big_field or chain_kernels_colour" 2>&1 | tail -4 | tee $O/c26_tests_small.txt
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "448" 2>&1 | tail -8 | tee $O/c26_tests_448.txt
B="--steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0"
timeout 900 python3 bench.py --workload 448V $B > $O/c26_bench_448V.json 2> $O/c26_bench_448V.err
tail -c 1500 $O/c26_bench_448V.json; tail -3 $O/c26_bench_448V.err
timeout 900 python3 bench.py --workload 512V $B > $O/c26_bench_512V.json 2> $O/c26_bench_512V.err
tail -c 1500 $O/c26_bench_512V.json; tail -5 $O/c26_bench_512V.err
