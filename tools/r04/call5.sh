#!/bin/bash
# round 4, call 5: PMC counters of k_line_sweep_pc against k_line_sweep_qpl on the 64 x 128 x 64 level shape (x-lines)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/pmc; rm -rf $O; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
run() {  # name, counters...
  local name=$1; shift
  EMG3D_PC_NL=2 timeout 100 rocprofv3 --pmc "$@" --output-format csv -d $O/pc_$name -- python3 tools/sweep_once.py 64 128 64 1 2 > $O/pc_$name.log 2>&1
  EMG3D_PC=0 timeout 100 rocprofv3 --pmc "$@" --output-format csv -d $O/qpl_$name -- python3 tools/sweep_once.py 64 128 64 1 2 > $O/qpl_$name.log 2>&1
  tail -1 $O/pc_$name.log $O/qpl_$name.log
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r04/pmc"
for d in sorted(glob.glob(O + "/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_line_sweep" not in k: continue
            acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
find $O -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
