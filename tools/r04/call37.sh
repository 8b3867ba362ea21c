#!/bin/bash
# round 4, call 37: HBM-side bytes of the mid-level kernels on 128 x 64 x 64, y-lines, coarse-level conditions (zeta read, dense
# right-hand side): k_line_sweep_tha (product choice), RS (EMG3D_THA=0), the scan kernel (EMG3D_THR=0) -- lab build,
# separate FETCH_SIZE / WRITE_SIZE passes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/c37; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so SWEEP_ONCE_COARSE=1
for v in tha rs qpl; do
  case $v in tha) export EMG3D_THA=3 EMG3D_THR=1;; rs) export EMG3D_THA=0 EMG3D_THR=1;; qpl) export EMG3D_THR=0;; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 tools/sweep_once.py 128 64 64 2 3 > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY' | tee gpurun_out/r04/c37_mid_level_traffic.txt
import csv, glob, collections
for v in ("tha", "rs", "qpl"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/r04/c37/{v}_{c}/*/*counter_collection.csv")
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "k_line_sweep" in r["Kernel_Name"]:
                vals[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, x in vals.items():
            tot.setdefault(k, {})[c] = sum(x) / len(x)
    for k, d in tot.items():
        b = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        blocks = 63 * 63 / 4 * 64      # one colour of the 128 x 64 x 64 grid's y-lines
        print(f"{v:4s} {k}: FETCH_SIZE {d['FETCH_SIZE']:.0f} KiB raw, WRITE_SIZE {d['WRITE_SIZE']:.0f} KiB -> {b/1e6:.1f} MB per launch, {b/ (127*63/4*64):.0f} B per block")
PY
