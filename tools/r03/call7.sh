#!/bin/bash
# round 3, GPU call 7: where a 256^3 V-cycle spends its time; the factor recurrence with one colour on the chip
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c7; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cycle256 -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 > $O/cycle256.log 2>&1
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so EMG3D_FACTOR_PER_COLOUR=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/factor256 -- python3 bench.py --mode sweep --workload 256V --no-cpu > $O/factor256.log 2>&1
find $O -type f ! -name '*kernel_stats.csv' ! -name '*.log' -delete
for f in $O/*.log; do tail -c 1200 $f > $f.t; mv $f.t $f; done
f=$(find $O/cycle256 -name '*kernel_stats.csv' | head -1); cut -c1-150 $f | head -22
f=$(find $O/factor256 -name '*kernel_stats.csv' | head -1); grep -i "factor\|sweep" $f | cut -c1-150
tail -c 600 $O/cycle256.log
