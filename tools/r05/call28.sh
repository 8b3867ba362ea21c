#!/bin/bash
# round 5, call 28: where does the 7 % spread of the 256^3 launch between processes come from? (tools/r05/bimodal.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
{
for p in 1 2 3; do echo "process $p"; timeout 600 python3 tools/r05/bimodal.py 256V 2>/dev/null; done
} | tee $O/c28_bimodal.txt
