#!/bin/bash
# round 5, call 17: the randomised parity sweeps on the final code (product and lab library; small, mid and long-line grids; handle reuse),
# then the cycle timelines of the final library
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 tests/tools/fuzz_parity.py 250 5001 > $O/c17_fuzz_default.txt 2>&1; echo "default rc=$?"; tail -2 $O/c17_fuzz_default.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so timeout 900 python3 tests/tools/fuzz_parity.py 250 5002 > $O/c17_fuzz_lab.txt 2>&1; echo "lab rc=$?"; tail -2 $O/c17_fuzz_lab.txt
FUZZ_SIZES=34,36,40,48,56,64,68,72,80 FUZZ_MAXCELLS=420000 FUZZ_MINMAX=64 timeout 1200 python3 tests/tools/fuzz_parity.py 30 5003 > $O/c17_fuzz_mid.txt 2>&1; echo "mid rc=$?"; tail -2 $O/c17_fuzz_mid.txt
FUZZ_SIZES=48,56,64,66,70,72,96,100,128 FUZZ_MAXCELLS=650000 FUZZ_MINMAX=96 timeout 1500 python3 tests/tools/fuzz_parity.py 20 5004 > $O/c17_fuzz_long.txt 2>&1; echo "long rc=$?"; tail -2 $O/c17_fuzz_long.txt
timeout 600 python3 tests/tools/fuzz_reuse.py 40 5005 > $O/c17_fuzz_reuse.txt 2>&1; echo "reuse rc=$?"; tail -2 $O/c17_fuzz_reuse.txt
for wl in 128F 256V; do
  D=/tmp/tr_$wl; rm -rf $D
  if [ $wl = 128F ]; then A="--steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-roofline"; N=6; else A="--workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline"; N=3; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py $A > /tmp/tr_$wl.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 tools/r05/gaps.py "$f" $N > $O/c17_gaps_$wl.txt 2>&1
  head -9 $O/c17_gaps_$wl.txt
done
