# A/B of kernel switches on bench.py (run on the GPU box): bash tools/ab_split.sh
run() { echo "$1 $2"; env $1 timeout 500 python bench.py --workload $2 --no-cpu --multi 0 --steps ${3:-4} --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['device_GB'])"; }
run A=1 128F 12
run EMG3D_SPLIT_MIN_CELLS=500000 128F 12
run EMG3D_SPLIT_MIN_CELLS=100000 128F 12
run EMG3D_QPL_MAX_NL=32 128F 12
run "EMG3D_QPL_MAX_NL=32 EMG3D_SPLIT_MIN_CELLS=500000" 128F 12
