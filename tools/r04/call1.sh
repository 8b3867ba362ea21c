#!/bin/bash
# round 4, call 1: parity of k_line_sweep_pc, then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "producer_chain" 2>&1 | tail -15 > gpurun_out/r04/c1_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/c1_bench.json 2> gpurun_out/r04/c1_bench.err
tail -c 1500 gpurun_out/r04/c1_pytest.txt
python tools/p.py gpurun_out/r04/c1_bench.json
