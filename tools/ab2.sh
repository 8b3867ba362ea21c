#!/bin/bash
# interleaved A/B of two environment settings on the cycle bench: A="X=1" B="X=2" N=3 bash tools/ab2.sh
for k in $(seq ${N:-3}); do
  for cfg in "$A" "$B"; do
    echo "$cfg: $(env $cfg timeout 300 python bench.py --workload ${WL:-128F} --no-cpu 2>&1 | tail -1 | python tools/p.py x | cut -d' ' -f2)"
  done
done
