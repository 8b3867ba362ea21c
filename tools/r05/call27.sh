#!/bin/bash
# round 5, call 27: k_line_sweep_qc<..., BIG>: two against three prefetch stages (lab knob EMG3D_Q_STAGES) at 512^3 and 448^3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(r.get("kernel"), round(r.get("launch_ms"),4), round(r.get("frac"),4), round(r.get("launch_ms_sparse_source") or 0,4), "cycle", round(d["ms_per_step"],2), d["rel_error_after"][-1])'
{
for w in 512V 448V; do for rep in 1 2; do for st in 3 2; do
  echo "Q_STAGES=$st $w: $(EMG3D_Q_STAGES=$st timeout 600 python3 bench.py --workload $w --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 2>/dev/null | python3 -c "$P")"
done; done; done
} 2>&1 | tee $O/c27_big_stages.txt
