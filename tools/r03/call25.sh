#!/bin/bash
# round 3, GPU call 25: k_line_sweep_lds -- parity (vs oracle, vs k_line_sweep_rp bit for bit) and timings
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c25; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
timeout 900 python3 -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "lds or (variant_matches and LDS)" > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
{
for lds in 0 1; do
  echo "EMG3D_LDS=$lds"
  EMG3D_LDS=$lds python3 tools/sweep_dirs.py 64 128 64
  EMG3D_LDS=$lds python3 tools/sweep_dirs.py 32 128 32
  EMG3D_LDS=$lds python3 tools/sweep_dirs.py 128 64 64
done
for rep in 1 2; do for lds in 0 1; do
  EMG3D_LDS=$lds timeout 600 python3 bench.py --workload 128F --no-cpu --multi 0 --no-dense --no-256 --no-tol --batch 0 --steps 12 --warmup 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('128F lds=$lds', round(d['ms_per_step'],3), d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
