#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for rep in 1 2; do for m2 in 64 32; do
  echo -n "QPL_M2=$m2 256V cycle: "; EMG3D_QPL_M2=$m2 timeout 300 python3 bench.py --workload 256V --steps 6 --warmup 2 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
  echo -n "QPL_M2=$m2 batch 8: "; EMG3D_QPL_M2=$m2 timeout 300 python3 tools/batch_cycle.py 128F 8 6 | tail -1
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c40_qpl_m2_256.txt
unset EMG3D_HIP_LIB
timeout 2600 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 | tee $O/c40_pytest.txt
