#!/bin/bash
# round 4, call 20: per-kernel statistics of the 128^3 F-cycle with k_line_sweep_thm<RS> on the mid levels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c20_cycle128 -- python3 bench.py --steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-dense > $O/c20_bench.txt 2>&1
f=$(ls $O/c20_cycle128/*/*kernel_stats.csv | head -1); cp $f $O/c20_cycle128_kernel_stats.csv
head -14 $O/c20_cycle128_kernel_stats.csv | cut -c1-150
