#!/bin/bash
# round 3, GPU call 26: longer randomised parity sweeps against the oracle with the final code -- the default path and the
# lab build with split copies forced on every level (fields at home in them) and the variants that exercise it
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c26; mkdir -p $O
timeout 900 python3 tests/tools/fuzz_parity.py 250 901 > $O/fuzz_default.txt 2>&1; echo "default rc=$?"; tail -4 $O/fuzz_default.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so EMG3D_SPLIT=1 timeout 900 python3 tests/tools/fuzz_parity.py 250 902 > $O/fuzz_split.txt 2>&1; echo "split rc=$?"; tail -4 $O/fuzz_split.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so EMG3D_SPLIT=1 EMG3D_QPL=0 EMG3D_Q=2 timeout 900 python3 tests/tools/fuzz_parity.py 150 903 > $O/fuzz_split_q.txt 2>&1; echo "split+q rc=$?"; tail -4 $O/fuzz_split_q.txt
timeout 600 python3 tests/tools/fuzz_reuse.py 60 904 > $O/fuzz_reuse.txt 2>&1; echo "reuse rc=$?"; tail -3 $O/fuzz_reuse.txt
