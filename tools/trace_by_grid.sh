#!/bin/bash
# kernel trace of a few 128^3 F-cycles grouped by (kernel, grid size): bash tools/trace_by_grid.sh TAG [ENV=V ...]
# (through gpurun; prints the table, keeps nothing but gpurun_out/<TAG>_bygrid.txt)
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=/tmp/tr_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 > /tmp/tr_$TAG.log 2>&1
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${TAG}_bygrid.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()), file=sys.stderr)
for r in rows:
    k = (r['Kernel_Name'].split('(')[0][:48], int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0) * int(r.get('Grid_Size_Y') or 1) * int(r.get('Grid_Size_Z') or 1) if 'Grid_Size' not in r else int(r['Grid_Size']))
    acc[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc[k][1] += 1
tot = sum(v[0] for v in acc.values())
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{k[0]:48s} grid {k[1]:9d} calls {n:5d} avg {t / n:8.1f} us total {t / 1e3:8.2f} ms {100 * t / tot:5.1f} %")
PY
tail -3 /tmp/tr_$TAG.log | cut -c1-200
grep ms_per_step /tmp/tr_$TAG.log | head -1 | cut -c1-200
cat gpurun_out/${TAG}_bygrid.txt
