#!/bin/bash
# round 3, GPU call 2: which kernel on the 64- and 32-block levels of the 128^3 F-cycle (scan vs two-sided chain, split copies)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c2; mkdir -p $O
run() { # tag env...
  local tag=$1; shift
  env "$@" timeout 300 python3 bench.py --steps 8 --warmup 3 --no-cpu --no-256 --no-tol --batch 0 --multi 0 > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json;d=json.load(open('$O/$tag.json'));print('$tag', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['cycles_to_1e-6'], d['rel_error_after'][-1])"
}
run base A=1
run qpl32 EMG3D_QPL_MAX_NL=32
run qpl16 EMG3D_QPL_MAX_NL=16
run qpl32_split EMG3D_QPL_MAX_NL=32 EMG3D_SPLIT_MIN_CELLS=500000
run qpl16_split EMG3D_QPL_MAX_NL=16 EMG3D_SPLIT_MIN_CELLS=120000
run qpl32_few0 EMG3D_QPL_MAX_NL=32 EMG3D_QPL_FEW=0
EMG3D_QPL_MAX_NL=32 EMG3D_LOG=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-256 --no-tol --batch 0 --multi 0 2>&1 >/dev/null | grep sweep | sort | uniq -c | sort -rn | head -20
