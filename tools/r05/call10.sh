#!/bin/bash
# round 5, call 10: the whole -m gpu suite on the library built from the committed sources (emg3d_amd/build_info.json), then the rocprofv3 collection (profiles/collect.sh r05)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/c10_tests.txt
bash profiles/collect.sh r05 > $O/c10_collect.log 2>&1
tail -3 $O/c10_collect.log
