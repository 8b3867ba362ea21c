#!/bin/bash
# round 4, call 35: randomised parity sweeps against the oracle on grids whose levels reach the mid-level kernel
# (k_line_sweep_tha: lines of 33..64 blocks, >= 1100 lines per colour), product library; plus the standard small-grid sweep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
FUZZ_SIZES=34,36,40,48,56,64,68,72,80 FUZZ_MAXCELLS=420000 FUZZ_MINMAX=64 timeout 1500 python3 tests/tools/fuzz_parity.py 40 1001 > $O/c35_fuzz_mid.txt 2>&1; echo "mid rc=$?"; tail -4 $O/c35_fuzz_mid.txt
timeout 900 python3 tests/tools/fuzz_parity.py 200 1002 > $O/c35_fuzz_default.txt 2>&1; echo "default rc=$?"; tail -4 $O/c35_fuzz_default.txt
timeout 600 python3 tests/tools/fuzz_reuse.py 40 1003 > $O/c35_fuzz_reuse.txt 2>&1; echo "reuse rc=$?"; tail -3 $O/c35_fuzz_reuse.txt
