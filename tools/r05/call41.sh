#!/bin/bash
# round 5, call 41: balanced lines per wave with C = one wave per SIMD: by grid size, product library against EMG3D_Q_BALANCE=0 (lab);
# the variants / kernel tests that touch the quad kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
{
echo "product library (balanced)"; timeout 1500 python3 tools/r05/size_scan.py 256 288 320 352 368 384 416 448 480 512 2>/dev/null
echo "lab, EMG3D_Q_BALANCE=0"; EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so EMG3D_Q_BALANCE=0 timeout 1500 python3 tools/r05/size_scan.py 256 288 384 448 2>/dev/null
} | tee $O/c41_balance.txt
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py tests/test_gpu_batch.py -q -x 2>&1 | tail -3 | tee $O/c41_tests.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "one_sweep or 448 or 256" 2>&1 | tail -3 | tee -a $O/c41_tests.txt
