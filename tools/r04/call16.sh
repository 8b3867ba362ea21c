#!/bin/bash
# round 4, call 16: k_line_sweep_thm<RS> after the counter-initialisation fix: tests (product library), repeated parity runs,
# isolated sweeps against the scan kernel, the bench cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_variants.py -q -m gpu -x 2>&1 | tail -5
{
for rep in 1 2 3; do timeout 250 python3 tools/r04/dbg_thr.py 64x48x44 32x66x100 24x130x34 64x80x80 25x47x49 | grep -v "e-1[0-9]"; echo "rep $rep done"; done
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for shp in "128 64 64" "128 32 32" "64 128 64" "32 128 32"; do
  for thr in 0 1; do echo -n "THR=$thr "; EMG3D_THR=$thr timeout 200 python3 tools/sweep_dirs.py $shp; done
  echo -n "THR=1 LPW=8 "; EMG3D_THR_LPW=8 timeout 200 python3 tools/sweep_dirs.py $shp
done
for rep in 1 2; do for thr in 0 1; do
  echo -n "THR=$thr 128F cycle: "; EMG3D_THR=$thr timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c16_thr.txt
