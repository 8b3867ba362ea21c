#!/bin/bash
# round 5, call 19: ring depth of k_line_sweep_tha (steps the helper waves may run ahead of the chain wave): 4 / 8 (default) / 12
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do for lib in lab thad4 thad12; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_$lib.so
  echo "$lib 128F: $(timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")   sweep 128x64x64 y: $(SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 128 64 64 2 20 2>/dev/null | tail -1)"
done; done
} 2>&1 | tee $O/c19_tha_ring.txt
