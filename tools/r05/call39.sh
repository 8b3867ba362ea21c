#!/bin/bash
# round 5, call 39: level-0 sweep efficiency against the grid size (tools/r05/size_scan.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python3 tools/r05/size_scan.py 256 364 368 372 384 388 392 404 406 408 320 288 2>/dev/null | tee $O/c39_size_scan.txt
