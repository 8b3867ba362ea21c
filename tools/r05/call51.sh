#!/bin/bash
# round 5, call 51: does a second wave per SIMD overlap in the two-sided kernel?  batched level-0 launches (2 / 8 systems: 2 / 8 waves per
# SIMD) with the ZS instantiation (260 registers: one wave per SIMD at a time) against the zeta-reading one (252: two fit; EMG3D_ZSEP=0)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(b["systems"], round(b["level0_sweep_launch_ms"]/b["systems"]*1e3,1), round(b["ms_per_cycle_per_system"],3)) for b in d["batched_sources"] if not b["batch_tune"]], d["roofline"]["kernel"], round(d["roofline"]["launch_ms"]*1e3,1))'
{
for rep in 1 2; do for z in 1 0; do
echo "EMG3D_ZSEP=$z: $(EMG3D_ZSEP=$z timeout 600 python3 bench.py --batch 2,8 --no-cpu --no-256 --no-tol --multi 0 2>/dev/null | python3 -c "$P")"
done; done
} | tee $O/c51_occupancy.txt
