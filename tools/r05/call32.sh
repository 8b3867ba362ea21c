#!/bin/bash
# round 5, call 32: TLB counters of the 256^3 level-0 launches in the fast and the slow placements of one process
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
rm -rf $O/c32_pmc $O/c32_pmc2
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $O/c32_pmc -- python3 tools/r05/bimodal.py 256V > $O/c32_run1.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $O/c32_pmc2 -- python3 tools/r05/bimodal.py 256V > $O/c32_run2.txt 2>&1
grep launch $O/c32_run1.txt; grep launch $O/c32_run2.txt
python3 - <<'P' | tee $O/c32_tlb.txt
import csv, glob, collections
for d in ("gpurun_out/r05/c32_pmc", "gpurun_out/r05/c32_pmc2"):
    cc = glob.glob(d + "/*/*counter_collection.csv"); kt = glob.glob(d + "/*/*kernel_trace.csv")
    if not cc or not kt: print("missing", d, cc, kt); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if "k_line_sweep_qc" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        if "k_line_sweep_qc" in r["Kernel_Name"]:
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows, key=int)
    # groups of consecutive dispatches: one handle = 3 + 30 sweeps x 4 launches = 132 launches
    n = 132
    print(d, len(ids), "launches")
    for g in range(0, len(ids), n):
        grp = ids[g:g + n]
        names = sorted(rows[grp[0]])
        avg = {k: sum(rows[i].get(k, 0) for i in grp) / len(grp) for k in names}
        du = [dur[i] for i in grp if i in dur]
        print("handle", g // n, "launch us %.1f" % (sum(du) / max(len(du), 1)), " ".join(f"{k}={v:.4g}" for k, v in avg.items()))
P
