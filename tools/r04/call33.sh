#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for rep in 1 2; do for v in 1 0; do
  echo -n "THR_NSYS1=$v batch 8: "; EMG3D_THR_NSYS1=$v timeout 300 python3 tools/batch_cycle.py 128F 8 6 | tail -1
  echo -n "THR_NSYS1=$v batch 3: "; EMG3D_THR_NSYS1=$v timeout 300 python3 tools/batch_cycle.py 128F 3 6 | tail -1
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c33_batch.txt
unset EMG3D_HIP_LIB
timeout 2600 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 | tee $O/c33_pytest.txt
