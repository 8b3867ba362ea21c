#!/bin/bash
# round 3, GPU call 1: p2p flag micro-benchmark, baseline bench line, tile map of the quad kernel (time + FETCH_SIZE)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c1; mkdir -p $O
timeout 120 ./tools/micro/p2p_flag > $O/p2p.txt 2>&1
timeout 600 python3 bench.py --no-cpu --no-tol --batch 0 > $O/bench_base.json 2> $O/bench_base.err
for t in 0 1; do
  EMG3D_Q_TILE=$t timeout 300 python3 bench.py --mode sweep --workload 256V --no-cpu > $O/sweep256_tile$t.json 2>> $O/sweep.err
  EMG3D_Q_TILE=$t timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_tile$t -- python3 bench.py --mode sweep --workload 256V --no-cpu > $O/fetch_tile$t.log 2>&1
  EMG3D_Q_TILE=$t timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_tile$t -- python3 bench.py --mode sweep --workload 256V --no-cpu > $O/write_tile$t.log 2>&1
done
EMG3D_Q_TILE=1 timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sweep" > $O/pytest_tile1.txt 2>&1
find $O -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1500 $f > $f.t; mv $f.t $f; done
python3 - <<'PY'
import csv, glob, collections
for t in (0, 1):
    for kind in ("fetch", "write"):
        fs = glob.glob(f"gpurun_out/r03c1/{kind}_tile{t}/**/*counter_collection.csv", recursive=True)
        if not fs: print(kind, t, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "k_line_sweep" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items(): print(kind, "tile", t, k, "launches", len(v), "mean", sum(v) / len(v))
PY
cat $O/p2p.txt
for t in 0 1; do python3 -c "
import json;d=json.load(open('$O/sweep256_tile$t.json'));r=d['roofline'];print('tile $t',r['kernel'],r['launch_ms'],r['sweep_ms'])"; done
python3 -c "
import json;d=json.load(open('$O/bench_base.json'));print(d['ms_per_step'],d['roofline']['launch_ms'],d['config_256V']['ms_per_cycle'],d['config_256V']['roofline']['launch_ms'])"
tail -3 $O/pytest_tile1.txt
