#!/bin/bash
# round 4, call 36: k_line_sweep_tha on 128-block lines of at most 2048 lines per colour (one round of workgroups):
# 256^3 V-cycle (level 2 = 128 x 128 x 64) and isolated sweeps on that shape
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for mx in 64 128; do echo -n "THA_MAX=$mx "; EMG3D_THA_MAX=$mx timeout 200 python3 tools/sweep_dirs.py 128 128 64; done
for mx in 64 128; do echo -n "THA_MAX=$mx "; EMG3D_THA_MAX=$mx timeout 200 python3 tools/sweep_dirs.py 128 64 64; done
for rep in 1 2; do for mx in 64 128; do
  echo -n "THA_MAX=$mx 256V: "; EMG3D_THA_MAX=$mx timeout 300 python3 bench.py --workload 256V --steps 6 --warmup 2 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
  echo -n "THA_MAX=$mx 128F: "; EMG3D_THA_MAX=$mx timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rel_error_after'][-1])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/c36_tha128.txt
