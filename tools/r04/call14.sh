#!/bin/bash
# round 4, call 14: k_line_sweep_thm<RS> (right-hand sides staged by helper waves): parity, isolated sweeps on the mid-level
# shapes against the scan kernel (EMG3D_THR=0, lab build), the bench cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "staged_right_hand" 2>&1 | tail -8
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "128 64 64" "128 32 32" "64 128 64" "32 128 32"; do
  for thr in 0 1; do echo -n "THR=$thr "; EMG3D_THR=$thr timeout 200 python3 tools/sweep_dirs.py $shp; done
  echo -n "THR=1 LPW=4 "; EMG3D_THR_LPW=4 timeout 200 python3 tools/sweep_dirs.py $shp
done
for rep in 1 2; do for thr in 0 1; do
  echo -n "THR=$thr 128F cycle: "; EMG3D_THR=$thr timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c14_thr.txt
