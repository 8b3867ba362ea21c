#!/bin/bash
# round 4, call 7: what the pipelined k_line_sweep_pc would do with perfectly coalesced, cached loads (every lane reads lane
# 0's addresses, no stores: timing only), with and without chain steps; SQ / TCP counters of the real kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32"; do
for nl in 2 4; do for dbg in 0 12 13; do echo "== $shp: pc NL=$nl dbg=$dbg"; EMG3D_PC_NL=$nl EMG3D_Q_TILE=$dbg timeout 200 python3 tools/sweep_dirs.py $shp; done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c7_uniform.txt
P=$O/pmc7; rm -rf $P; mkdir -p $P
run() { local name=$1; shift
  EMG3D_PC_NL=4 timeout 100 rocprofv3 --pmc "$@" --output-format csv -d $P/pc_$name -- python3 tools/sweep_once.py 64 128 64 1 2 > $P/pc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
run fetch FETCH_SIZE
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r04/pmc7"
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
find $P -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
