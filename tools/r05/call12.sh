#!/bin/bash
# round 5, call 12: prefetch depth of the quad kernel at 384^3 (36.6 k lines per colour = 2.2 waves per SIMD at 16 lines per wave)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],2), "dense", round(1e3*r["launch_ms"],1), "dipole", round(1e3*r["launch_ms_sparse_source"],1), r["kernel"])'
{
for rep in 1 2; do for st in 0 3; do
  echo "Q_STAGES=$st 384V: $(EMG3D_Q_STAGES=$st timeout 600 python3 bench.py --workload 384V --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 2>/dev/null | python3 -c "$P")"
done; done
} 2>&1 | tee $O/c12_qstages_384.txt
