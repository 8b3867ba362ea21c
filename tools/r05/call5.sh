#!/bin/bash
# round 5, call 5: what-if of a compact mirrored factor (11 instead of 15 / 14 numbers per block and pass) in k_line_sweep_thm / _tha:
# a build that only SKIPS the loads (-DEMG3D_WHATIF_CF: wrong results, timing and traffic only) against the lab build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],3), "dense", round(1e3*r["launch_ms"],2), "dipole", round(1e3*r["launch_ms_sparse_source"],2), r["kernel"])'
{
for rep in 1 2 3; do for lib in lab cfx; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_$lib.so
  echo "$lib 128F: $(timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 2>/dev/null | python3 -c "$P")"
done; done
for lib in lab cfx; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_$lib.so
  echo "$lib tha 128x64x64 y-lines, coarse-level conditions: $(SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 128 64 64 2 20 2>/dev/null | tail -1)"
  echo "$lib tha 128x64x64 z-lines: $(SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 128 64 64 3 20 2>/dev/null | tail -1)"
done
} 2>&1 | tee $O/c5_whatif_cf.txt
