#!/bin/bash
# round 4, call 10: the whole GPU suite after the pruning of the lab build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 2000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -25 > $O/c10_pytest.txt
tail -6 $O/c10_pytest.txt
