#!/bin/bash
# round 5, call 13: per-lane launch descriptors of the scan kernel (LineArgs::qd, smooth_qpl.hpp DM = 2): parity, then A/B on the 128^3 F-cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_batch.py tests/test_gpu_variants.py -q -m gpu -x 2>&1 | tail -4 | tee $O/c13_tests.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do for v in "EMG3D_QDESC=0" "EMG3D_QDESC=1" "EMG3D_QDESC_MAX=9000" "EMG3D_QDESC_MAX=70000"; do
  echo "$v 128F: $(env $v timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-roofline 2>/dev/null | python3 -c "$P")"
done; done
echo "== stamps with descriptors, 128 x 4 x 4 y-lines"; SWEEP_ONCE_COARSE=1 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py 128 4 4 2 2 2>&1 | grep -v amdgpu.ids | tail -6
echo "== 128 x 16 x 16 y-lines"; SWEEP_ONCE_COARSE=1 EMG3D_Q_TILE=512 timeout 120 python3 tools/sweep_once.py 128 16 16 2 2 2>&1 | grep -v amdgpu.ids | tail -6
} 2>&1 | tee $O/c13_qdesc_ab.txt
