#!/bin/bash
# round 5, call 37: the -m gpu suite with per-test durations (which tests carry the 750 s)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=40 2>&1 | tail -60 | tee $O/c37_durations.txt
