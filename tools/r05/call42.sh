#!/bin/bash
# round 5, call 42: colours of 8192 ... 16383 lines through the balanced 16-line instantiation (one round of waves) against the 8-line
# instantiation (EMG3D_Q_BALANCE=0, lab): level-0 sweeps at 192^3 ... 250^3; the 256^3 V-cycle (its level 1 takes the new path)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for b in 0 1; do echo "EMG3D_Q_BALANCE=$b"; EMG3D_Q_BALANCE=$b timeout 1500 python3 tools/r05/size_scan.py 184 192 200 216 224 240 250 2>/dev/null; done
for rep in 1 2; do for b in 0 1; do echo "EMG3D_Q_BALANCE=$b 256V cycle: $(EMG3D_Q_BALANCE=$b timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense --no-roofline 2>/dev/null | python3 -c "$P")"; done; done
} | tee $O/c42_balance_mid.txt
unset EMG3D_HIP_LIB
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py tests/test_gpu_batch.py -q -x 2>&1 | tail -3 | tee $O/c42_tests.txt
