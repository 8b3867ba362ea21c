#!/bin/bash
# round 4, call 45: scan kernel prologue (shifts for the power-of-two divisions, selects): parity, then the cycle against the
# previous build (emg3d_amd/libemg3d_hip_prev.so = commit 8810431)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -4
{
for rep in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib 128F: "; timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['rel_error_after'][-1])"
done; done
for lib in prev new; do
  if [ $lib = prev ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_prev.so; else export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip.so; fi
  echo -n "$lib lex: "; timeout 300 python3 bench.py --ordering lex --steps 1 --warmup 1 --no-cpu --no-256 --no-tol --multi 0 --batch 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['rel_error_after'][-1])"
  echo -n "$lib batch 8: "; timeout 300 python3 tools/batch_cycle.py 128F 8 6 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c45_qpl_prologue_ab.txt
