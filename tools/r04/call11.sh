#!/bin/bash
# round 4, call 11: batched level-0 sweeps (8 systems): two systems per workgroup (EMG3D_THM_PAIRSYS) against the default;
# HBM / L1->L2 counters of both; bit-identity of the batched solve; the pointer-snapshot test
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for ps in 0 1; do for n in 2 4 8; do for d in 1 3; do
  EMG3D_THM_PAIRSYS=$ps timeout 200 python3 tools/batch_sweep.py 128F $n $d 5 | sed "s/^/pairsys $ps: /"
done; done; done
for ps in 0 1; do EMG3D_THM_PAIRSYS=$ps timeout 300 python3 tools/batch_cycle.py 128F 8 6 | sed "s/^/pairsys $ps: /"; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c11_pairsys.txt
P=$O/pmc11; rm -rf $P; mkdir -p $P
for ps in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
    n=$(echo $c | cut -c1-5)
    EMG3D_THM_PAIRSYS=$ps timeout 120 rocprofv3 --pmc $c --output-format csv -d $P/ps${ps}_$n -- python3 tools/batch_sweep.py 128F 8 3 2 > $P/ps${ps}_$n.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r04/pmc11"
for d in sorted(glob.glob(O + "/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_line_sweep" not in k: continue
            acc[k.split("(")[0][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
find $P -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
unset EMG3D_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_batch.py -q -m gpu -x -k "snapshot or bitwise" 2>&1 | tail -4
# level 1 of the 256^3 V-cycle (256 x 128 x 128: 8192 lines x 128 blocks per colour, zeta read, dense source): bytes per block
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P=gpurun_out/r04/pmc11b; rm -rf $P; mkdir -p $P
for c in FETCH_SIZE WRITE_SIZE; do
  SWEEP_ONCE_COARSE=1 timeout 200 rocprofv3 --pmc $c --output-format csv -d $P/l1_$c -- python3 tools/sweep_once.py 256 128 128 2 2 > $P/l1_$c.log 2>&1
  tail -1 $P/l1_$c.log
done
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r04/pmc11b"
for d in sorted(glob.glob(O + "/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_line_sweep" not in k: continue
            acc[k.split("(")[0][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
find $P -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
