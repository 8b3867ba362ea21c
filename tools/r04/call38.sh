#!/bin/bash
# round 4, call 38: randomised parity sweeps on grids with an axis of 96..128 cells (k_line_sweep_tha on lines of 65..128 blocks)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
FUZZ_SIZES=48,56,64,66,70,72,96,100,128 FUZZ_MAXCELLS=650000 FUZZ_MINMAX=96 timeout 2400 python3 tests/tools/fuzz_parity.py 30 1004 > $O/c38_fuzz_long.txt 2>&1; echo "long rc=$?"; tail -4 $O/c38_fuzz_long.txt
