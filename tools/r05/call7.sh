#!/bin/bash
# round 5, call 7: level 1 of the 256^3 V-cycle (8192 lines x 128 blocks per colour) with the two-sided kernel instead of the quad kernel
# (lab knobs EMG3D_Q_MIN_LINES / EMG3D_TWIST_MAX = 8193), alternating on one box; then the cycle's timeline by kernel class
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do
  echo "default        256V: $(timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense --mode cycle 2>/dev/null | python3 -c "$P")"
  echo "thm on level 1 256V: $(EMG3D_Q_MIN_LINES=8193 EMG3D_TWIST_MAX=8193 timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense 2>/dev/null | python3 -c "$P")"
done
} 2>&1 | tee $O/c7_level1_thm.txt
for v in default thm1; do
  D=/tmp/tr_$v; rm -rf $D
  if [ $v = thm1 ]; then export EMG3D_Q_MIN_LINES=8193 EMG3D_TWIST_MAX=8193; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense > /tmp/tr_$v.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 tools/r05/gaps.py "$f" 3 > $O/c7_gaps_256V_$v.txt 2>&1
  head -24 $O/c7_gaps_256V_$v.txt
done
