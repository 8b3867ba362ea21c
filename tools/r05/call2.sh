#!/bin/bash
# round 5, call 2: cache-policy A/B of the level-0 kernels (lab builds with -DEMG3D_NT=<mask>: 1 factor loads, 2 parked z, 4 result
# stores, 8 source loads, 16 neighbour loads), cycle + isolated sweeps (dense / dipole) at 128^3 and 256^3; and the quad kernel with
# 2 prefetch stages at 16 lines per wave only (EMG3D_Q_STAGES=23)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],3), "dense", round(1e3*r["launch_ms"],2), "dipole", round(1e3*r["launch_ms_sparse_source"],2), r["kernel"], d["rel_error_after"][-1])'
{
for rep in 1 2; do for m in 0 1 2 4 3 7 8 16; do
  export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_nt$m.so
  echo -n "nt$m 128F: "; timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 | python3 -c "$P"
  echo -n "nt$m 256V: "; timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 | python3 -c "$P"
done; done
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_nt0.so
for rep in 1 2 3; do for st in 3 23; do
  echo -n "Q_STAGES=$st 256V: "; EMG3D_Q_STAGES=$st timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 | python3 -c "$P"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c2_nt_ab.txt
