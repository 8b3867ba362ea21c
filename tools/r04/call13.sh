#!/bin/bash
# round 4, call 13: argument bursts in residual / restriction / prolongation / point sweep too: bench cycle against the library
# before the bursts (libemg3d_hip_old.so), alternating; 256^3 V-cycle; the parity suites that cover these kernels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
NEW=$PWD/emg3d_amd/libemg3d_hip.so; OLD=$PWD/emg3d_amd/libemg3d_hip_old.so
{
for rep in 1 2 3; do for lib in OLD NEW; do
  echo -n "$lib 128F cycle: "; EMG3D_HIP_LIB=${!lib} timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
for lib in OLD NEW; do
  echo -n "$lib 256V cycle: "; EMG3D_HIP_LIB=${!lib} timeout 300 python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"
done
for lib in OLD NEW; do echo -n "$lib "; EMG3D_HIP_LIB=${!lib} timeout 300 python3 tools/batch_cycle.py 128F 8 6; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c13_burst_all.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -3
