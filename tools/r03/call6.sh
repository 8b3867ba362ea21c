#!/bin/bash
# round 3, GPU call 6: the whole GPU suite (product library; tests/test_gpu_variants.py on the lab build)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c6; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu --durations=15 > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
