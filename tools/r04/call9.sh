#!/bin/bash
# round 4, call 9: the new GPU tests (epsilon_r, 384^3 sweep parity, factor-offset boundary, pointer snapshot, 8 ranks) + a bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_shard.py tests/test_gpu_fullsize.py -q -m gpu -x \
   -k "epsilon or model_paths or eight_ranks or 384 or offset_boundary or snapshot" 2>&1 | tail -15 > $O/c9_pytest.txt
tail -5 $O/c9_pytest.txt
timeout 600 python bench.py --steps 12 --warmup 3 > $O/c9_bench.json 2> $O/c9_bench.err; tail -3 $O/c9_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/c9_bench.json"))
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "reference_order_lex")})
print({k: r.get(k) for k in ("kernel", "frac", "frac_sparse_source", "launch_ms", "launch_ms_sparse_source", "formulation_floor_frac", "traffic_stale", "traffic")})
print(d["cycle_algorithmic"]["frac"], d["cycle_algorithmic"]["frac_executed"], d["code"])
c = d["config_256V"]; r = c["roofline"]
print(c["ms_per_cycle"], {k: r.get(k) for k in ("kernel", "frac", "frac_sparse_source", "launch_ms", "formulation_floor_frac")})
PY
