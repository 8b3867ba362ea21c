#!/bin/bash
# round 5, call 54: level 1 of the 256^3 V-cycle (8192 lines of 128 blocks per colour): lines per wave 8 (one round of 1024 half-filled waves)
# against 10 / 12 / 16 (820 / 683 / 512 fuller waves) -- experiment library with a forced runtime value (throwaway patch of q_balanced_lpw)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/tools/r05/exp/libemg3d_hip_exp.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2; do for l in 8 10 12 16; do
echo "lines per wave on level 1 = $l: 256V cycle $(EMG3D_Q_RT_LPW=$l timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense --no-roofline 2>/dev/null | python3 -c "$P")   sweep 256x128x128 y: $(EMG3D_Q_RT_LPW=$l SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 256 128 128 2 10 2>/dev/null | tail -1)"
done; done
} | tee $O/c54_level1_lpw.txt
