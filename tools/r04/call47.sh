#!/bin/bash
# round 4, call 47: the randomised parity sweeps again on the final code (after the prologue / index-arithmetic changes of R4.11-R4.14):
# small grids incl. the lab variants (one-sided / two-sided / split / lexicographic), mid and long-line grids, handle reuse
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python3 tests/tools/fuzz_parity.py 250 2001 > $O/c47_fuzz_default.txt 2>&1; echo "default rc=$?"; tail -2 $O/c47_fuzz_default.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so timeout 900 python3 tests/tools/fuzz_parity.py 250 2002 > $O/c47_fuzz_lab.txt 2>&1; echo "lab rc=$?"; tail -2 $O/c47_fuzz_lab.txt
FUZZ_SIZES=34,36,40,48,56,64,68,72,80 FUZZ_MAXCELLS=420000 FUZZ_MINMAX=64 timeout 1200 python3 tests/tools/fuzz_parity.py 30 2003 > $O/c47_fuzz_mid.txt 2>&1; echo "mid rc=$?"; tail -2 $O/c47_fuzz_mid.txt
FUZZ_SIZES=48,56,64,66,70,72,96,100,128 FUZZ_MAXCELLS=650000 FUZZ_MINMAX=96 timeout 1500 python3 tests/tools/fuzz_parity.py 20 2004 > $O/c47_fuzz_long.txt 2>&1; echo "long rc=$?"; tail -2 $O/c47_fuzz_long.txt
timeout 600 python3 tests/tools/fuzz_reuse.py 40 2005 > $O/c47_fuzz_reuse.txt 2>&1; echo "reuse rc=$?"; tail -2 $O/c47_fuzz_reuse.txt
