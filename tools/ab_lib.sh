# A/B of two builds of the library (EMG3D_HIP_LIB) on isolated sweeps and the bench cycle
for lib in "$@"; do
  echo "== $lib"
  for shp in "32 128 32" "16 128 16" "8 128 8" "4 128 4" "32 32 32"; do EMG3D_HIP_LIB=$PWD/$lib python tools/sweep_dirs.py $shp 2>/dev/null | tail -1; done
  for i in 1 2; do EMG3D_HIP_LIB=$PWD/$lib python bench.py --no-cpu --multi 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cycle', round(d['ms_per_step'],3))"; done
done
