#!/bin/bash
# round 3, GPU call 14: the whole GPU suite on the product library, then on the lab build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c14; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_product.txt 2>&1
tail -14 $O/pytest_product.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so timeout 2400 python3 -m pytest tests -q -m gpu -k "not vcycle_vs_oracle and not one_iteration_vs_oracle and not two_cycles" > $O/pytest_lab.txt 2>&1
tail -4 $O/pytest_lab.txt
