# A/B of kernel switches on the split-layout level-0 sweeps (run on the GPU box)
run() { echo "$1 $2"; env $1 timeout 500 python bench.py --workload $2 --no-cpu --multi 0 --steps ${3:-4} --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['device_GB'])"; }
run A=1 256V
run EMG3D_LPW=12 256V
run EMG3D_LPW=8 256V
