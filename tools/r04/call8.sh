#!/bin/bash
# round 4, call 8: pipelined k_line_sweep_pc with a tick barrier that leaves the prefetched loads in flight
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_variants.py -q -m gpu -k "producer_chain" 2>&1 | tail -4 > $O/c8_pytest.txt
tail -2 $O/c8_pytest.txt
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for shp in "64 128 64" "32 128 32" "16 128 16"; do
echo "== $shp: qpl"; EMG3D_PC=0 timeout 200 python3 tools/sweep_dirs.py $shp
for nl in 2 4; do for dbg in 0 12 13; do echo "== $shp: pc NL=$nl dbg=$dbg"; EMG3D_PC_NL=$nl EMG3D_Q_TILE=$dbg timeout 200 python3 tools/sweep_dirs.py $shp; done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c8_barrier.txt
