#!/bin/bash
# round 6, call F: GPU tests touched by the chain form / placement, then the default bench line
mkdir -p gpurun_out/r06/f
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_batch.py tests/test_gpu_logging.py -x -q 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/r06/f/bench_128F.json 2> gpurun_out/r06/f/bench_128F.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/f/bench_128F.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "vs_profile", d["roofline"].get("vs_profile"))
print("roofline_256V", {k: d["roofline_256V"].get(k) for k in ("launch_ms", "frac", "vs_profile", "placement_mode", "ms_per_cycle", "Mcells_per_s", "target_met")})
print("placement", d["roofline_256V"].get("placement"))
print("parity", {k: v for k, v in d["parity"].items() if k.startswith("max")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
