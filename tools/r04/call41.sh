#!/bin/bash
# round 4, call 41: have other launch-selection thresholds moved now that the coarse launches are shorter?  128^3 F-cycle, lab build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
b() { timeout 300 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['rel_error_after'][-1])"; }
{
for rep in 1 2; do
echo -n "default              "; b
echo -n "XT_MIN=1000          "; EMG3D_XT_MIN=1000 b
echo -n "XT_MIN=100000        "; EMG3D_XT_MIN=100000 b
echo -n "QPL_FEW=2100         "; EMG3D_QPL_FEW=2100 b
echo -n "QPL_FEW=512          "; EMG3D_QPL_FEW=512 b
echo -n "THR_MIN_LINES=1000   "; EMG3D_THR_MIN_LINES=1000 b
echo -n "QPL_MAX_NL=32        "; EMG3D_QPL_MAX_NL=32 b
echo -n "TH_LPW=12            "; EMG3D_TH_LPW=12 b
echo -n "THR_MIN=32 lines 900 "; EMG3D_THR_MIN=32 EMG3D_THR_MIN_LINES=900 b
done
} 2>&1 | grep -v amdgpu.ids | tee $O/c41_thresholds.txt
