#!/bin/bash
# round 5, call 49: the randomised parity sweeps on the final code (new seeds), incl. a set of WIDE grids whose level 0 has 4600 ... 10 000
# lines per colour in some direction (two-sided kernel at 12 lines per pair, quad kernel at 9 ... 10 lines per wave: HISTORY R5.19)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 tests/tools/fuzz_parity.py 200 6001 > $O/c49_fuzz_default.txt 2>&1; echo "default rc=$?"; tail -1 $O/c49_fuzz_default.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so timeout 900 python3 tests/tools/fuzz_parity.py 200 6002 > $O/c49_fuzz_lab.txt 2>&1; echo "lab rc=$?"; tail -1 $O/c49_fuzz_lab.txt
FUZZ_SIZES=34,36,40,48,56,64,68,72,80 FUZZ_MAXCELLS=420000 FUZZ_MINMAX=64 timeout 1200 python3 tests/tools/fuzz_parity.py 24 6003 > $O/c49_fuzz_mid.txt 2>&1; echo "mid rc=$?"; tail -1 $O/c49_fuzz_mid.txt
FUZZ_SIZES=34,40,136,144,152,184,192,200 FUZZ_MAXCELLS=1400000 FUZZ_MINMAX=136 timeout 2400 python3 tests/tools/fuzz_parity.py 24 6004 > $O/c49_fuzz_wide.txt 2>&1; echo "wide rc=$?"; tail -1 $O/c49_fuzz_wide.txt
timeout 600 python3 tests/tools/fuzz_reuse.py 40 6005 > $O/c49_fuzz_reuse.txt 2>&1; echo "reuse rc=$?"; tail -1 $O/c49_fuzz_reuse.txt
