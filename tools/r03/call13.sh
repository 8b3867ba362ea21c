#!/bin/bash
# round 3, GPU call 13: level 1 of the 256^3 V-cycle (8192 lines x 128 blocks): quad kernel 8 / 16 lines per wave, two-sided kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c13; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
run() { # tag env...
  local tag=$1; shift
  env "$@" timeout 400 python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --no-dense --batch 0 > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json;d=json.load(open('$O/$tag.json'));print('$tag', round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['launch_ms'],4))"
}
run base A=1
run lpw16 EMG3D_Q_LPW=16
run thm_l1 EMG3D_TWIST_MAX=8193 EMG3D_Q_MIN_LINES=8193
run base2 A=1
