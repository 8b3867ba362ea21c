"""Turn the raw rocprofv3 output merged into gpurun_out/ by profiles/collect.sh
into the committed summaries under profiles/ (+ traffic.json used by bench.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def one(pattern):
    fs = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


KERNELS = {}


TIMED_LAUNCHES = 360    # bench.py --mode sweep: SWEEP_SAMPLES x 3 directions x (1 warm-up + 2 timed sweeps) x 4 colour launches


def counters(name, kernel="k_line_sweep_"):
    """Mean counter values per launch of the DOMINANT line-sweep kernel of a `bench.py --mode sweep` run: the instantiation with the
    largest counter total, and of it only the last TIMED_LAUNCHES dispatches -- the set-up in front of the sweeps launches the same kernel
    on candidate blocks (placement) and small instantiations of the scan kernel (launch descriptors), which are not the launches
    the roofline is priced on."""
    f = one(f"{tag}_{name}/*/*counter_collection.csv")
    if not f:
        return {}
    rows = collections.defaultdict(list)        # (kernel, counter) -> [(dispatch, value)]
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            rows[(r["Kernel_Name"], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    if not rows:
        return {}
    totals = collections.defaultdict(float)
    for (k, c), v in rows.items():
        totals[k] += sum(x for _, x in v)
    dom = max(totals, key=totals.get)
    # "void k_line_sweep_qc<c128, 3, 16>(LineArgs<c128>)" -> the name bench.py reports: "k_line_sweep_qc<c128,3,16>"
    KERNELS[name] = dom.replace("void ", "").split("(")[0].replace(", ", ",")
    out = {}
    for (k, c), v in rows.items():
        if k == dom:
            v = [x for _, x in sorted(v)][-TIMED_LAUNCHES:]
            out[c] = sum(v) / len(v)
    return out


traffic = {}
lines = []
for wl, suffix in (("128F", "128"), ("256V", "256")):
    st = one(f"{tag}_sweep{suffix}/*/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(PROF, f"{tag}_sweep_{wl}_kernel_stats.csv"))
    st = one(f"{tag}_sweep{suffix}d/*/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(PROF, f"{tag}_sweep_{wl}_dense_kernel_stats.csv"))
    fe = counters(f"fetch{suffix}").get("FETCH_SIZE")
    wr = counters(f"write{suffix}").get("WRITE_SIZE")
    fed = counters(f"fetch{suffix}d").get("FETCH_SIZE")
    wrd = counters(f"write{suffix}d").get("WRITE_SIZE")
    if fe and wr:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
        # exactly half of the bytes of wide (16 B/lane) reads -> doubled; WRITE_SIZE is exact.
        hbm = (2 * fe + wr) * 1024
        traffic[wl] = {"hbm_bytes_per_launch": hbm, "kernel": KERNELS.get(f"fetch{suffix}"),
                       "ratio_to_algorithmic": hbm / (200.0 * {"128F": 128, "256V": 256}[wl] ** 3 / 4),
                       "FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB": wr,
                       "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB) per k_line_sweep_* launch, mean over launches"}
        lines.append(f"{wl}: FETCH_SIZE {fe:.0f} KiB (raw), WRITE_SIZE {wr:.0f} KiB -> HBM bytes/launch {hbm/1e6:.0f} MB")
        if fed and wrd:
            hbmd = (2 * fed + wrd) * 1024
            traffic[wl].update({"hbm_bytes_per_launch_dense_source": hbmd,
                                "ratio_to_algorithmic_dense_source": hbmd / (200.0 * {"128F": 128, "256V": 256}[wl] ** 3 / 4),
                                "FETCH_SIZE_KiB_raw_dense_source": fed, "WRITE_SIZE_KiB_dense_source": wrd})
            lines.append(f"{wl} dense source: FETCH_SIZE {fed:.0f} KiB (raw), WRITE_SIZE {wrd:.0f} KiB -> {hbmd/1e6:.0f} MB")
for wl, suffix in (("128F", "128"), ("256V", "256")):
    for kind, sfx in (("", ""), ("_dense", "d")):
        tj = os.path.join(OUT, f"{tag}_sweep{suffix}{sfx}_timed.json")
        if os.path.exists(tj) and os.path.getsize(tj) > 10:
            t = json.load(open(tj))
            # the profiled process's own HIP-event time of the same launches (its bench line): the two timers side by side
            lj = os.path.join(OUT, f"{tag}_sweep{suffix}{sfx}_line.json")
            try:
                own = json.load(open(lj))["roofline"]
                t["hip_event_ms_same_process"] = own["launch_ms"]
                t["profiler_over_hip_events"] = t["average_ms"] / own["launch_ms"]
                t["placement_same_process"] = {k: (v.get("tries"), v.get("kept")) for k, v in
                                               (own.get("placement") or {}).get("per_working_copy", {}).items()}
            except (OSError, ValueError, KeyError):
                pass
            json.dump(t, open(os.path.join(PROF, f"{tag}_sweep_{wl}{kind}_timed_launches.json"), "w"))
for wl, name in (("128F", "cycle128"), ("256V", "cycle256")):
    tl = os.path.join(OUT, f"{tag}_{name}_timeline.txt")
    if os.path.exists(tl):
        shutil.copy(tl, os.path.join(PROF, f"{tag}_cycle_{wl}_timeline.txt"))


def by_kernel(name):
    """mean counter value per launch, by kernel, of a --pmc run over whole cycles"""
    f = one(f"{tag}_{name}/*/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fc, wc = by_kernel("fetch256c"), by_kernel("write256c")
if fc and wc:
    with open(os.path.join(PROF, f"{tag}_cycle_256V_traffic_by_kernel.txt"), "w") as f:
        f.write("# HBM traffic per launch of every kernel of the 256^3 V-cycle (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of\n"
                "# `bench.py --workload 256V --steps 3 --warmup 3 --no-roofline`, incl. set-up launches; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> MB)\n")
        f.write(f"{'kernel':66s} {'launches':>8s} {'fetch MB':>10s} {'write MB':>10s} {'HBM MB':>10s}\n")
        for k in sorted(fc, key=lambda k: -(2 * fc[k][0] + wc.get(k, (0, 0))[0]) * fc[k][1]):
            fe, n = fc[k]
            wr = wc.get(k, (0.0, 0))[0]
            f.write(f"{k[:66]:66s} {n:8d} {2 * fe * 1024 / 1e6:10.1f} {wr * 1024 / 1e6:10.1f} {(2 * fe + wr) * 1024 / 1e6:10.1f}\n")

st = one(f"{tag}_cycle128/*/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(PROF, f"{tag}_cycle_128F_kernel_stats.csv"))
st = one(f"{tag}_cycle256/*/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(PROF, f"{tag}_cycle_256V_kernel_stats.csv"))
st = one(f"{tag}_bench/*/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(PROF, f"{tag}_bench_kernel_stats.csv"))
st = one(f"{tag}_batch8/*/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(PROF, f"{tag}_batch8_128F_kernel_stats.csv"))
sq = counters("sq128")
if sq:
    with open(os.path.join(PROF, f"{tag}_sweep_128F_sq_counters.json"), "w") as f:
        json.dump(sq, f, indent=1)
for name in ("bench_128F", "bench_256V", "bench_384V", "bench_448V", "bench_512V", "bench_128F_lex", "bench_128F_multi3", "bench_128F_batch"):
    src = os.path.join(OUT, f"{tag}_{name}.json")
    if os.path.exists(src) and os.path.getsize(src) > 10:
        shutil.copy(src, os.path.join(PROF, f"{tag}_{name}.json"))
if traffic:
    import subprocess
    box = one(f"{tag}_box.txt")
    bi = {}
    bif = one(f"{tag}_build_info.json")
    if bif:
        bi = json.load(open(bif))
    traffic["source"] = {
        "collected_by": f"profiles/collect.sh {tag}: the same gpurun call (box, library) as the committed {tag}_bench_*.json lines",
        "box": open(box).read().strip() if box else None,
        # what the library ON THAT BOX was built from (emg3d_amd/build_info.json travels with the snapshot); bench.py sets
        # roofline.traffic_stale when the library it runs was built from other sources
        "sources_commit": bi.get("sources_commit"), "sources_dirty": bi.get("dirty"), "built": bi.get("built"),
    }
    with open(os.path.join(PROF, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
# The bench lines of this collection were printed on the GPU box BEFORE these summaries existed (bench.py read the previous
# round's traffic.json there and flagged it stale).  Attach what the SAME gpurun call measured -- PMC traffic, rocprof averages of
# the same library on the same box -- to the committed copies, and say so.
if traffic:
    sys.path.insert(0, ROOT)
    import bench as _bench
    for name in ("bench_128F", "bench_256V"):
        dst = os.path.join(PROF, f"{tag}_{name}.json")
        if not os.path.exists(dst):
            continue
        # the line as the GPU box printed it stays beside the annotated copy
        shutil.copy(dst, os.path.join(PROF, f"{tag}_{name}.raw.json"))
        line = json.load(open(dst))
        todo = [(line.get("roofline"), "128F" if name == "bench_128F" else "256V")]
        if "config_256V" in line:
            todo.append((line["config_256V"].get("roofline"), "256V"))
            todo.append(((line.get("roofline") or {}).get("at_256V"), "256V"))
        # the library that printed the line (line["code"]) against the one the counters were collected on (traffic["source"]):
        # only the same sources, clean at build time on both sides, make the counters this line's own
        code, src = line.get("code") or {}, traffic.get("source") or {}
        same_build = bool(code.get("sources_commit")) and code.get("sources_commit") == src.get("sources_commit") and \
            not code.get("sources_dirty") and not src.get("sources_dirty")
        for r, wl in todo:
            ent = traffic.get(wl)
            if not r or not ent or not ent.get("kernel") or not r.get("kernel") or \
                    not ent["kernel"].startswith(r["kernel"].rstrip(">")):
                continue
            dense = r.get("launch_ms_sparse_source") is not None
            r["traffic"] = ent.get("hbm_bytes_per_launch_dense_source") if dense else ent["hbm_bytes_per_launch"]
            r["traffic_sparse_source"] = ent["hbm_bytes_per_launch"] if dense else None
            r["traffic_stale"] = not same_build
            r["traffic_source"] = {"file": "profiles/traffic.json", "kernel": ent["kernel"], "measured_in_this_run": False,
                                   "attached_by": "profiles/summarise.py: collected in the same gpurun call as this line", **traffic["source"]}
            if r["traffic"]:
                r["traffic_rate_GBs"] = r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9
                if "hbm_stream" in line:
                    r["traffic_rate_vs_copy"] = r["traffic_rate_GBs"] / line["hbm_stream"]["copy_GBs"]
            r["rocprof_average"] = _bench._rocprof_average_ms(r["kernel"], f"sweep_{wl}_dense" if dense else "bench")
            r["rocprof_average_sparse_source"] = _bench._rocprof_average_ms(r["kernel"], "bench") if dense else None
            # ... and the self-check against THIS collection's profile (on the box the line was checked against the previous one)
            ra = r["rocprof_average"]
            if ra and ra.get("average_ms"):
                r["vs_profile"] = r["launch_ms"] / ra["average_ms"]
                r.pop("placement_mode", None)
                if not 0.97 <= r["vs_profile"] <= 1.03:
                    r["placement_mode"] = ("faster than the committed profile's process" if r["vs_profile"] < 1 else
                                           "slower than the committed profile's process") + " (physical placement / box: HISTORY R5.18, R6.1)"
        # the copies of the 256^3 roofline inside the default line follow their source
        r2 = line.get("config_256V", {}).get("roofline")
        if r2:
            for tgt in (line.get("roofline", {}).get("at_256V"), line.get("roofline_256V")):
                if tgt is not None:
                    for k in ("traffic", "traffic_stale", "traffic_sparse_source", "rocprof_average", "rocprof_average_sparse_source", "vs_profile",
                              "traffic_rate_GBs"):
                        if k in r2:
                            tgt[k] = r2[k]
                    tgt.pop("placement_mode", None)
                    if "placement_mode" in r2:
                        tgt["placement_mode"] = r2["placement_mode"]
        json.dump(line, open(dst, "w"))
print("\n".join(lines))
print("SQ:", sq)
