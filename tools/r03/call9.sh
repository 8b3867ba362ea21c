#!/bin/bash
# round 3, GPU call 9: two-sided quad kernel (lab: k_line_sweep_qm) on the 64-block level of the 128^3 F-cycle
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c9; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
run() { # tag env...
  local tag=$1; shift
  env "$@" timeout 300 python3 bench.py --steps 8 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --multi 0 > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json;d=json.load(open('$O/$tag.json'));print('$tag', round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['launch_ms'],4), d['cycles_to_1e-6'])"
}
run base A=1
run qm_all32 EMG3D_QM=1 EMG3D_QPL_MAX_NL=32
run qm_all32_l2 EMG3D_QM=1 EMG3D_QPL_MAX_NL=32 EMG3D_QM_LPW=2
run qm_all32_l4 EMG3D_QM=1 EMG3D_QPL_MAX_NL=32 EMG3D_QM_LPW=4
run qm_all32_l1 EMG3D_QM=1 EMG3D_QPL_MAX_NL=32 EMG3D_QM_LPW=1
run qm_all16_l2 EMG3D_QM=1 EMG3D_QPL_MAX_NL=16 EMG3D_QM_LPW=2
EMG3D_QM=1 EMG3D_QPL_MAX_NL=32 EMG3D_LOG=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-256 --no-tol --batch 0 --multi 0 2>&1 >/dev/null | grep sweep | sort | uniq -c | sort -rn | head -8
