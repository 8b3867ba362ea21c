#!/bin/bash
# round 3, GPU call 16: coarse levels at home in the x-split copies too -- parity and A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c16; mkdir -p $O
LAB=$PWD/emg3d_amd/libemg3d_hip_lab.so
timeout 1500 python3 -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "home" > $O/pytest_home.txt 2>&1
tail -12 $O/pytest_home.txt
for wl in 256V; do for rep in 1 2 3; do for home in 1 0; do
  EMG3D_HOME=$home EMG3D_HIP_LIB=$LAB timeout 600 python3 bench.py --workload $wl --no-cpu --multi 0 --no-dense --steps 12 --warmup 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$wl home=$home', round(d['ms_per_step'],3))"
done; done; done 2>&1 | tee $O/ab.txt
timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=5 > $O/pytest_product.txt 2>&1
tail -8 $O/pytest_product.txt
