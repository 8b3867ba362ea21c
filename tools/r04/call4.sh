#!/bin/bash
# round 4, call 4: parity of k_line_sweep_pc (5-lane producer mapping), isolated sweeps on the mid-level shapes, phases, cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "producer_chain" 2>&1 | tail -15 > $O/c4_pytest.txt
tail -3 $O/c4_pytest.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32" "16 128 16"; do
  echo "== $shp: qpl (EMG3D_PC=0)"; EMG3D_PC=0 timeout 300 python3 tools/sweep_dirs.py $shp
  for nl in 1 2 4; do echo "== $shp: pc NL=$nl"; EMG3D_PC_NL=$nl timeout 300 python3 tools/sweep_dirs.py $shp; done
done
for dbg in 1 5; do echo "== 64 128 64: pc NL=2 dbg=$dbg"; EMG3D_PC_NL=2 EMG3D_Q_TILE=$dbg timeout 300 python3 tools/sweep_dirs.py 64 128 64; done
echo "== 128 128 128: PC=0 / PC=1"
EMG3D_PC=0 timeout 300 python3 tools/sweep_dirs.py 128 128 128
timeout 300 python3 tools/sweep_dirs.py 128 128 128
} 2>&1 | grep -v amdgpu.ids | tee $O/c4_mid_level.txt
