#!/bin/bash
# round 4, call 21: RS with the final selection rule (33..64 blocks, >= 1100 lines, 8 lines per workgroup, no transposed copies)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "staged_right" 2>&1 | tail -5
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "64 64 128" "40 80 80"; do
  for thr in 0 1; do echo -n "THR=$thr "; EMG3D_THR=$thr timeout 200 python3 tools/sweep_dirs.py $shp; done
done
for rep in 1 2; do for thr in 0 1; do
  echo -n "THR=$thr 128F cycle: "; EMG3D_THR=$thr timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
echo -n "256V THR=0: "; EMG3D_THR=0 timeout 300 python3 bench.py --workload 256V --steps 6 --warmup 2 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
echo -n "256V THR=1: "; EMG3D_THR=1 timeout 300 python3 bench.py --workload 256V --steps 6 --warmup 2 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
} 2>&1 | grep -v amdgpu.ids | tee $O/c21_thr.txt
