#!/bin/bash
# round 5, call 24: the whole -m gpu suite on the final tree
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/c24_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee -a $O/c24_tests.txt
