#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 300 python3 tools/solve_breakdown.py 128F 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/c16_solve_breakdown.txt
EMG3D_LOG_SETUP=1 timeout 300 python3 tools/setup_time.py 128F 2>&1 | grep -v amdgpu.ids | tail -40 | tee -a $O/c16_solve_breakdown.txt
