#!/bin/bash
# k_residual: A/B of the block map + PMC passes (run through gpurun; summaries land in gpurun_out/resid/)
set -u
O=gpurun_out/resid; mkdir -p $O
timeout 600 python3 tools/ab_residual.py > $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for x in 0/1 0/4 2/4; do
  xt=$(echo $x | tr / _)
  for c in "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY TD_TD_BUSY GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp_${xt}_$tag -o out -- python3 $R/tools/ab_residual.py 128F only=$x > /tmp/rp_${xt}_$tag.log 2>&1
    f=$(find /tmp/rp_${xt}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$x" "$c" >> $R/$O/pmc.txt <<'PY'
import csv, sys, collections
f, x, c = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'k_residual' in r['Kernel_Name']:
        k = (r['Kernel_Name'][:40], r['Counter_Name'], r.get('Grid_Size', ''))
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"RES_XCD/KZ={x} {k[0]} grid {k[2]} {k[1]}: {v / n:.4g} per launch ({n} launches)")
PY
  done
done
cat $R/$O/ab.txt $R/$O/pmc.txt
