#!/bin/bash
# round 5, call 50: 32-block lines with >= 1100 lines per colour (two rounds of the scan kernel's one-line waves) through the affine kernel
# (lab knob EMG3D_THA_MIN=32): level 3 of the 256^3 V-cycle; the 128^3 F-cycle must not change (its 32-block level has 1024 lines)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), d["rel_error_after"][-1])'
{
for rep in 1 2 3; do for m in 33 32; do
echo "EMG3D_THA_MIN=$m 256V: $(EMG3D_THA_MIN=$m timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense --no-roofline 2>/dev/null | python3 -c "$P")"
done; done
for m in 33 32; do echo "EMG3D_THA_MIN=$m 128F: $(EMG3D_THA_MIN=$m timeout 300 python3 bench.py --steps 8 --warmup 3 --no-cpu --no-tol --no-256 --multi 0 --batch 0 --no-dense --no-roofline 2>/dev/null | python3 -c "$P")"; done
for m in 33 32; do echo "EMG3D_THA_MIN=$m sweep 256x32x32 y: $(EMG3D_THA_MIN=$m SWEEP_ONCE_COARSE=1 timeout 120 python3 tools/sweep_once.py 256 32 32 2 20 2>/dev/null | tail -1)"; done
} | tee $O/c50_tha32.txt
