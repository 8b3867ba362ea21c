#!/bin/bash
# round 3, GPU call 4: LIFO with branch-free phases -- parity + time
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c4; mkdir -p $O
for t in 1 0 1 0; do
  EMG3D_THM_LIFO=$t timeout 300 python3 bench.py --mode sweep --no-cpu > $O/sweep128_lifo$t.json 2>> $O/sweep.err
  python3 -c "
import json;d=json.load(open('$O/sweep128_lifo$t.json'));r=d['roofline'];print('lifo $t',r['kernel'],r['launch_ms'],r['sweep_ms'])"
done
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sweep or two_cycles" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
