#!/bin/bash
# HISTORY R6.4: what a result store fused into the level-0 sweep -- into the x<->y transposed working copy -- costs (lab what-if,
# EMG3D_Q_TILE=1024: y- / z-line sweeps store every result a second time at transposed-layout addresses), against the transposition kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06/t
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
# (the x-lines' working copy must exist: the cycle mode prepares all three pairs before the sweeps are timed)
for rep in 1 2; do for t in 0 1024; do
  EMG3D_Q_TILE=$t EMG3D_PLACE_TRIES=0 timeout 300 python bench.py --workload 256V --steps 2 --warmup 1 --no-cpu --no-tol --batch 0 > gpurun_out/r06/t/tile${t}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json, glob
for t in (0, 1024):
    for f in sorted(glob.glob(f"gpurun_out/r06/t/tile{t}_*.json")):
        r = json.load(open(f))["roofline"]
        print(f"EMG3D_Q_TILE={t}: dense-source sweep ms (4 launches) x {r['sweep_ms']['x']:.3f} y {r['sweep_ms']['y']:.3f} z {r['sweep_ms']['z']:.3f}")
PY
for t in 0 1024; do for c in FETCH_SIZE WRITE_SIZE; do
  EMG3D_Q_TILE=$t EMG3D_PLACE_TRIES=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r06/t/pmc_${t}_$c -- python3 bench.py --workload 256V --steps 1 --warmup 1 --no-cpu --no-tol --batch 0 > /dev/null 2>&1
  python3 - gpurun_out/r06/t/pmc_${t}_$c $t $c <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_line_sweep_qc<c128, 2, 16" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v = v[-48:]          # the isolated dense-source sweeps come last
    print(f"EMG3D_Q_TILE={sys.argv[2]} {k}: mean over the last {len(v)} level-0 launches {sum(v)/len(v)*1024/1e6:.1f} MB (raw KiB x 1024)")
PY
  rm -rf gpurun_out/r06/t/pmc_${t}_$c
done; done
