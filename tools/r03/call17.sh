#!/bin/bash
# round 3, GPU call 17: coarse levels at home in the x-split copies -- parity after the smooth_point fix
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c17; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_variants.py -q -m gpu -k "home" > $O/pytest_home.txt 2>&1
tail -6 $O/pytest_home.txt
timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=5 > $O/pytest_product.txt 2>&1
tail -8 $O/pytest_product.txt
EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so timeout 2400 python3 -m pytest tests -q -m gpu -k "not vcycle_vs_oracle and not one_iteration_vs_oracle and not two_cycles" > $O/pytest_lab.txt 2>&1
tail -4 $O/pytest_lab.txt
