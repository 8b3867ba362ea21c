#!/bin/bash
# round 4, call 12: k_line_sweep_qpl with the kernel arguments loaded in one burst and all global loads of a launch issued
# together (libemg3d_hip.so) against the version before (libemg3d_hip_old.so), alternating, on the coarse-level shapes and the
# bench cycle; parity of the scan kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
NEW=$PWD/emg3d_amd/libemg3d_hip.so; OLD=$PWD/emg3d_amd/libemg3d_hip_old.so
{
for rep in 1 2; do
for shp in "128 4 4" "128 8 8" "128 16 16" "128 32 32" "128 64 64" "4 128 4" "16 16 128"; do
  for lib in OLD NEW; do echo -n "$lib "; EMG3D_HIP_LIB=${!lib} timeout 200 python3 tools/sweep_dirs.py $shp; done
done; done
for rep in 1 2 3; do for lib in OLD NEW; do
  echo -n "$lib 128F cycle: "; EMG3D_HIP_LIB=${!lib} timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c12_burst.txt
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py tests/test_gpu_solver.py -q -m gpu -x -k "scan_kernels or gauss or regression or solves_16" 2>&1 | tail -3
