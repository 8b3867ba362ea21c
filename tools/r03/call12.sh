#!/bin/bash
# round 3, GPU call 12: source-free lines skip the source loads: bit-identity tests, parity at full size, time and traffic
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c12; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "source_free or zeta_from" > $O/pytest1.txt 2>&1
tail -4 $O/pytest1.txt
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_solver.py tests/test_gpu_krylov.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest2.txt 2>&1
tail -4 $O/pytest2.txt
timeout 600 python3 bench.py --no-cpu --no-tol --batch 0 > $O/bench.json 2> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench.json'));r=d['roofline'];c=d['config_256V'];q=c['roofline']
print('128F', d['ms_per_step'], r['launch_ms'], r['launch_ms_dense_source'], r['frac']); print('256V', c['ms_per_cycle'], q['launch_ms'], q['launch_ms_dense_source'], q['frac'])"
for wl in 128F 256V; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${c}_$wl -- python3 bench.py --mode sweep --workload $wl --no-cpu > $O/${c}_$wl.log 2>&1
  done
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1500 $f > $f.t; mv $f.t $f; done
python3 - <<'PY'
import csv, glob, collections
for wl in ("128F", "256V"):
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r03c12/{kind}_{wl}/**/*counter_collection.csv", recursive=True)
        if not fs: print(kind, wl, "no csv"); continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "k_line_sweep" in r["Kernel_Name"]]
        v = [float(r["Counter_Value"]) for r in rows]
        # the first third of the launches: the workload's dipole; the dense-source launches follow
        print(wl, kind, "launches", len(v), "mean all", sum(v) / len(v), "first 16", sum(v[:16]) / 16, "last 16", sum(v[-16:]) / 16)
PY
