#!/bin/bash
# round 5, call 4: timeline of the timed cycles (busy time and gaps by kernel class), 128^3 F-cycle and 256^3 V-cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for wl in 128F 256V; do
  D=/tmp/tr_$wl; rm -rf $D
  if [ $wl = 128F ]; then A="--steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-dense"; N=6; else A="--workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense"; N=3; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py $A > /tmp/tr_$wl.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 tools/r05/gaps.py "$f" $N > $O/c4_gaps_$wl.txt 2>&1
  head -12 $O/c4_gaps_$wl.txt
done
