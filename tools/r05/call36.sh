#!/bin/bash
# round 5, call 36: finer map of the copy / two-read rate against the distance of the streams (tools/micro/placement2.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 tools/micro/placement2.py 2>/dev/null > $O/c36_placement_map.txt; tail -5 $O/c36_placement_map.txt
