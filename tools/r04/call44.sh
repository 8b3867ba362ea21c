#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 2600 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 | tee $O/c44_pytest.txt
bash profiles/collect.sh r04
