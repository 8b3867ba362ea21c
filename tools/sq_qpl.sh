#!/bin/bash
# SQ counters of the quad-per-block kernel on a 64x128x64 grid (the level-1 shape of the 128^3 F-cycle)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sqq
rm -rf $O ${O}b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O -- python3 tools/sweep_dirs.py 64 128 64 > $O.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d ${O}b -- python3 tools/sweep_dirs.py 64 128 64 > ${O}b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("$O", "${O}b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "qpl" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:40] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
