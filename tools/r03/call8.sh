#!/bin/bash
# round 3, GPU call 8: new tests (cgs on the device, interp3d boundary modes), then call 7's profiles
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c8; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_receivers.py tests/test_gpu_source.py tests/test_gpu_kernels.py tests/test_hfield.py tests/test_gpu_gradient.py -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
bash tools/r03/call7.sh
