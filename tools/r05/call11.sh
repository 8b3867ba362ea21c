#!/bin/bash
# round 5, call 11: the cycle-only kernel statistics again, without the isolated sweeps of the roofline object in the profile (--no-roofline)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; TAG=r05
run() { local name=$1; shift; rm -rf $OUT/${TAG}_${name}; rocprofv3 "$@" > $OUT/${TAG}_${name}.log 2>&1
  find $OUT/${TAG}_${name} -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
  tail -c 2000 $OUT/${TAG}_${name}.log > $OUT/${TAG}_${name}.log.tail; mv $OUT/${TAG}_${name}.log.tail $OUT/${TAG}_${name}.log; }
run cycle128 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cycle128 -- python3 bench.py --steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-roofline
run cycle256 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cycle256 -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline
tail -c 300 $OUT/${TAG}_cycle256.log
