#!/bin/bash
# round 5, call 1: the new colour-vs-reference parity tests + shard test on the GPU; baseline numbers of this box;
# A/B of the quad kernel's register prefetch depth (lab knob EMG3D_Q_STAGES: 3 = 322 VGPRs with AGPR copies, 2 = 210, none) at 256^3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shard.py -q -m gpu -x 2>&1 | tail -4 | tee $O/c1_tests.txt
{
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
for rep in 1 2 3; do for st in 3 2; do
  echo -n "Q_STAGES=$st 256V sweep dense: "; EMG3D_Q_STAGES=$st timeout 300 python3 bench.py --mode sweep --source dense --workload 256V --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r.get('kernel'), r.get('launch_ms'), r.get('frac'))"
  echo -n "Q_STAGES=$st 256V sweep dipole: "; EMG3D_Q_STAGES=$st timeout 300 python3 bench.py --mode sweep --workload 256V --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r.get('kernel'), r.get('launch_ms'), r.get('frac'))"
done; done
for rep in 1 2; do for st in 3 2; do
  echo -n "Q_STAGES=$st 256V cycle: "; EMG3D_Q_STAGES=$st timeout 300 python3 bench.py --workload 256V --steps 4 --warmup 3 --no-cpu --no-tol --batch 0 --no-dense | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['rel_error_after'][-1])"
done; done
unset EMG3D_HIP_LIB
} 2>&1 | grep -v amdgpu.ids | tee $O/c1_qstages_ab.txt
timeout 600 python3 bench.py > $O/c1_bench_128F.json 2> $O/c1_bench_128F.err; tail -c 600 $O/c1_bench_128F.json
