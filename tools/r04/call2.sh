#!/bin/bash
# round 4, call 2: parity of k_line_sweep_pc; isolated sweeps on the mid-level shapes of the 128^3 F-cycle, pc against qpl
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "producer_chain" 2>&1 | tail -15 > $O/c2_pytest.txt
tail -3 $O/c2_pytest.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32" "16 128 16" "128 64 64" "128 32 32"; do
  echo "== $shp: qpl (EMG3D_PC=0)"; EMG3D_PC=0 timeout 300 python3 tools/sweep_dirs.py $shp
  for nl in 1 2 4; do echo "== $shp: pc NL=$nl"; EMG3D_PC_NL=$nl timeout 300 python3 tools/sweep_dirs.py $shp; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c2_mid_level.txt
