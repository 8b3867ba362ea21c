#!/bin/bash
# SQ counters of the sweep kernel on one workload: WL=64F bash tools/sq.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${WL:-64F}
O=gpurun_out/sq_$WL
rm -rf $O ${O}b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O -- python3 bench.py --mode sweep --workload $WL --no-cpu > $O.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d ${O}b -- python3 bench.py --mode sweep --workload $WL --no-cpu > ${O}b.log 2>&1
tail -2 $O.log ${O}b.log
ls $O ${O}b
