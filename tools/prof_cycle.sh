#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_cycle_now
rm -rf $O
rocprofv3 --kernel-trace --stats -d $O -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu ${BENCH_ARGS} > $O.log 2>&1
tail -1 $O.log | cut -c1-300
