#!/bin/bash
# round 3, GPU call 10: zeta formed from the width vectors in k_line_sweep_qc (ZS): parity, time, counted traffic
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c10; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sweep" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for t in 1 0 1 0; do
  EMG3D_ZSEP=$t timeout 300 python3 bench.py --mode sweep --workload 256V --no-cpu > $O/sweep256_zs$t.json 2>> $O/sweep.err
  python3 -c "
import json;d=json.load(open('$O/sweep256_zs$t.json'));r=d['roofline'];print('zsep $t',r['kernel'],r['launch_ms'],r['sweep_ms'])"
done
for t in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EMG3D_ZSEP=$t timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${c}_zs$t -- python3 bench.py --mode sweep --workload 256V --no-cpu > $O/${c}_zs$t.log 2>&1
  done
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1500 $f > $f.t; mv $f.t $f; done
python3 - <<'PY'
import csv, glob, collections
for t in (1, 0):
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r03c10/{kind}_zs{t}/**/*counter_collection.csv", recursive=True)
        if not fs: print(kind, t, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "k_line_sweep" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items(): print("zsep", t, k, "launches", len(v), "mean", sum(v) / len(v))
PY
unset EMG3D_HIP_LIB
timeout 600 python3 bench.py --no-cpu --no-tol --batch 0 > $O/bench.json 2> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['roofline']['launch_ms'],d['config_256V']['ms_per_cycle'],d['config_256V']['roofline']['launch_ms'],d['config_256V']['roofline']['frac'])"
