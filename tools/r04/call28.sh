#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
{
timeout 300 python3 tools/r04/dbg_thr.py 128x70x66 100x68x72 | grep -v "dir [23].*e-1[0-9]"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "128 or boundary" 2>&1 | tail -5
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for rep in 1 2; do for mx in 64 128; do
  echo -n "THA_MAX=$mx 128F cycle: "; EMG3D_THA_MAX=$mx timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1], d['roofline']['kernel'], d['roofline'].get('launch_ms'))"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c28_tha_l0.txt
