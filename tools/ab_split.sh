run() { echo "$1 $2"; env $1 timeout 500 python bench.py --workload $2 --no-cpu --multi 0 --steps ${3:-4} --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['device_GB'])"; }
run EMG3D_SPLIT=0 256V
run EMG3D_SPLIT=2 256V
run EMG3D_SPLIT_MIN_CELLS=4000000 256V
run EMG3D_SPLIT_MIN_CELLS=2000000 256V
run EMG3D_SPLIT=0 128F 12
run EMG3D_SPLIT_MIN_CELLS=2000000 128F 12
run EMG3D_SPLIT_MIN_CELLS=1000000 128F 12
