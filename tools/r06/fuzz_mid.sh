#!/bin/bash
# randomised parity on MID-SIZE grids (34 ... 128 cells per axis: the two-sided, affine and scan kernels in their product regimes) on the final code
mkdir -p gpurun_out/r06/fuzz
FUZZ_SIZES=34,36,40,48,56,64,72,80,96,128 FUZZ_MAXCELLS=900000 FUZZ_MINMAX=40 timeout 2400 python tests/tools/fuzz_parity.py ${1:-40} ${2:-66} > gpurun_out/r06/fuzz/parity_mid.txt 2>&1
echo "rc $?"; tail -5 gpurun_out/r06/fuzz/parity_mid.txt
