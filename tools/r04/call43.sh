#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib 128F: "; timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), d['rel_error_after'][-1], r['kernel'], r.get('launch_ms'), r.get('launch_ms_sparse_source'))"
done; done
for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib 256V: "; timeout 300 python3 bench.py --workload 256V --steps 6 --warmup 2 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['rel_error_after'][-1])"
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c43_prologue_ab.txt
