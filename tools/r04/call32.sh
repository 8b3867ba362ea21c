#!/bin/bash
# full GPU suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 2600 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 | tee $O/c32_pytest.txt
