#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 64 32" "32 64 64"; do
  echo -n "default      "; timeout 200 python3 tools/sweep_dirs.py $shp
  echo -n "tha nL>=32   "; EMG3D_THR_MIN=32 EMG3D_THR_MIN_LINES=1000 timeout 200 python3 tools/sweep_dirs.py $shp
done
for rep in 1 2; do
  echo -n "default: "; timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
  echo -n "tha nL>=32, >=1000 lines: "; EMG3D_THR_MIN=32 EMG3D_THR_MIN_LINES=1000 timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c27_tha32.txt
