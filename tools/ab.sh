#!/bin/bash
# usage: VAR=NAME VALS="a b" WLS="128F 64F" MODE=sweep|cycle bash tools/ab.sh   (A/B of one env switch)
for wl in ${WLS:-128F}; do
  for v in $VALS; do
    if [ "${MODE:-sweep}" = sweep ]; then args="--mode sweep"; else args=""; fi
    echo "$wl $VAR=$v: $(env $VAR=$v timeout 300 python bench.py $args --workload $wl --no-cpu 2>&1 | tail -1 | python tools/p.py x)"
  done
done
