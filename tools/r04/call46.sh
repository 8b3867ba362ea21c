#!/bin/bash
# round 4, call 46: running store offsets in k_line_sweep_thm: parity at full size, then the cycle against the previous build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_batch.py tests/test_gpu_variants.py -q -m gpu -x 2>&1 | tail -4
{
for rep in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib 128F: "; timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), d['rel_error_after'][-1], r.get('launch_ms'))"
done; done
for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib batch 8: "; timeout 300 python3 tools/batch_cycle.py 128F 8 6 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c46_thm_stores_ab.txt
