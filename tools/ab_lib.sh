# A/B of builds of the library (EMG3D_HIP_LIB) on bench.py: bash tools/ab_lib.sh <workload> lib1 lib2 ...
wl=$1; shift
for rep in 1 2; do for lib in "$@"; do
  EMG3D_HIP_LIB=$PWD/$lib python bench.py --workload $wl --no-cpu --multi 0 --steps ${STEPS:-8} --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))"
done; done
