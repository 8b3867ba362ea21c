#!/bin/bash
# round 3, GPU call 3: LDS LIFO of k_line_sweep_thm: parity at 128^3, launch time and counted traffic with / without
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c3; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sweep or two_cycles" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for t in 1 0; do
  EMG3D_THM_LIFO=$t timeout 300 python3 bench.py --mode sweep --no-cpu > $O/sweep128_lifo$t.json 2>> $O/sweep.err
  for c in FETCH_SIZE WRITE_SIZE; do
    EMG3D_THM_LIFO=$t timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${c}_lifo$t -- python3 bench.py --mode sweep --no-cpu > $O/${c}_lifo$t.log 2>&1
  done
  python3 -c "
import json;d=json.load(open('$O/sweep128_lifo$t.json'));r=d['roofline'];print('lifo $t',r['kernel'],r['launch_ms'],r['sweep_ms'])"
done
for t in 1 0; do
EMG3D_THM_LIFO=$t timeout 300 python3 bench.py --steps 8 --warmup 3 --no-cpu --no-tol --batch 0 --multi 0 > $O/bench_lifo$t.json 2>> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench_lifo$t.json'));print('lifo $t cycle',d['ms_per_step'],d['roofline']['launch_ms'],d['config_256V']['ms_per_cycle'],d['config_256V']['roofline']['launch_ms'])"
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1500 $f > $f.t; mv $f.t $f; done
python3 - <<'PY'
import csv, glob, collections
for t in (1, 0):
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r03c3/{kind}_lifo{t}/**/*counter_collection.csv", recursive=True)
        if not fs: print(kind, t, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "k_line_sweep" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items(): print("lifo", t, k, "launches", len(v), "mean", sum(v) / len(v))
PY
