#!/bin/bash
# round 5, call 43: level-0 sweep efficiency by size in the two-sided kernel's regime (< 8192 lines per colour) and across the 8192 border
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python3 tools/r05/size_scan.py 96 112 128 136 144 152 160 168 176 180 182 184 2>/dev/null | tee $O/c43_size_scan_thm.txt
