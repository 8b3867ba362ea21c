#!/bin/bash
# round 5, call 52: what the driver runs at round end, on the final tree: the -m gpu suite, smoke(), the default bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | tee $O/c52_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $O/c52_smoke.txt
timeout 900 python bench.py > $O/c52_bench.json 2> $O/c52_bench.err; python3 -c "
import json; d=json.loads(open('$O/c52_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'], round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'at256', round(r['at_256V']['frac'],4), 'stale', r['traffic_stale'], 'cpu', round(d['cpu_baseline']['value'],3), d['code'])"
