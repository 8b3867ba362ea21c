#!/bin/bash
# round 5, call 35: copy rate against the distance of source and destination (tools/micro/placement.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
{ timeout 600 python3 tools/micro/placement.py 1 2>/dev/null; echo; timeout 600 python3 tools/micro/placement.py 1 2>/dev/null | tail -8; } | tee $O/c35_placement_copy.txt
