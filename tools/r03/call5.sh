#!/bin/bash
# round 3, GPU call 5: compact factor (k_line_sweep_qc): parity, launch time and counted traffic against k_line_sweep_q
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c5; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sweep" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for t in 1 0 1 0; do
  EMG3D_QC=$t timeout 300 python3 bench.py --mode sweep --workload 256V --no-cpu > $O/sweep256_qc$t.json 2>> $O/sweep.err
  python3 -c "
import json;d=json.load(open('$O/sweep256_qc$t.json'));r=d['roofline'];print('qc $t',r['kernel'],r['launch_ms'],r['sweep_ms'])"
done
for t in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EMG3D_QC=$t timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${c}_qc$t -- python3 bench.py --mode sweep --workload 256V --no-cpu > $O/${c}_qc$t.log 2>&1
  done
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1500 $f > $f.t; mv $f.t $f; done
python3 - <<'PY'
import csv, glob, collections
for t in (1, 0):
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r03c5/{kind}_qc{t}/**/*counter_collection.csv", recursive=True)
        if not fs: print(kind, t, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "k_line_sweep" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items(): print("qc", t, k, "launches", len(v), "mean", sum(v) / len(v))
PY
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py -x -q -m gpu > $O/pytest2.txt 2>&1
tail -5 $O/pytest2.txt
