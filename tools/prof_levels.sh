#!/bin/bash
# per-(level, direction) time of the sweep launches of eager cycles: launch log + kernel trace
# (summarise with: python tools/levels_table.py)
cd /tmp; export TMPDIR=/tmp
export EMG3D_GRAPH=0 EMG3D_LOG=1
O=$GRAFT_REPO_ROOT/gpurun_out/prof_levels
rm -rf $O
rocprofv3 --kernel-trace -d $O -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu ${BENCH_ARGS} > $O.out 2> $O.err
grep -c "^\[sweep\]" $O.err
