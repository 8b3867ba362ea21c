#!/bin/bash
# round 4, call 6: parity of the pipelined k_line_sweep_pc (chain wave + three producer waves), isolated sweeps, counters
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "producer_chain" 2>&1 | tail -15 > $O/c6_pytest.txt
tail -3 $O/c6_pytest.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32" "16 128 16"; do
  echo "== $shp: qpl (EMG3D_PC=0)"; EMG3D_PC=0 timeout 200 python3 tools/sweep_dirs.py $shp
  for nl in 1 2 4; do echo "== $shp: pc NL=$nl"; EMG3D_PC_NL=$nl timeout 200 python3 tools/sweep_dirs.py $shp; done
done
for dbg in 1 5; do echo "== 64 128 64: pc NL=4 dbg=$dbg"; EMG3D_PC_NL=4 EMG3D_Q_TILE=$dbg timeout 200 python3 tools/sweep_dirs.py 64 128 64; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c6_mid_level.txt
