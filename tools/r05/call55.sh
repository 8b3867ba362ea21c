#!/bin/bash
# round 5, call 55: where a 384^3 and a 448^3 V-cycle spend their time (kernel trace of the timed cycles, tools/r05/gaps.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for wl in 384V 448V; do
  D=/tmp/tr_$wl; rm -rf $D
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $wl --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 --no-roofline > /tmp/tr_$wl.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 tools/r05/gaps.py "$f" 3 > $O/c55_gaps_$wl.txt 2>&1
  head -24 $O/c55_gaps_$wl.txt
done
