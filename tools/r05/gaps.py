"""Timeline of the timed cycles of a `rocprofv3 --kernel-trace` run of bench.py: per cycle the sum of kernel durations, the sum of the
gaps between consecutive kernels, and both by kernel class (name + grid size).  python tools/r05/gaps.py <kernel_trace.csv> [ncycles]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:44],
             int(r.get('Grid_Size') or 0)) for r in rows)
# the timed cycles are the LAST ncyc end-of-cycle norms: find the k_sum_sqrt launches (one per cycle) from the end
idx = [i for i, e in enumerate(ev) if e[2].startswith('k_sum_sqrt')]
# the isolated roofline sweeps and the residual timing come after the cycles: take the longest run of norm launches that are one
# cycle apart (same number of kernels in between, within 2 %)
best = None
for j in range(len(idx) - 1, ncyc - 1, -1):
    seg = idx[j - ncyc:j + 1]
    d = [b - a for a, b in zip(seg, seg[1:])]
    if min(d) > 100 and best is None:
        best = seg
        break
if best is None:
    print("no run of", ncyc, "cycles found"); sys.exit(1)
tot_k = tot_g = 0.0
byk = collections.defaultdict(lambda: [0, 0.0, 0.0])
for a, b in zip(best, best[1:]):
    cyc = ev[a + 1:b + 1]
    dur = sum(e[1] - e[0] for e in cyc) / 1e3
    gaps = 0.0
    prev_end = ev[a][1]
    for e in cyc:
        g = max(0, e[0] - prev_end) / 1e3
        gaps += g
        k = byk[(e[2], e[3])]
        k[0] += 1; k[1] += (e[1] - e[0]) / 1e3; k[2] += g
        prev_end = e[1]
    print(f"cycle: {len(cyc)} kernels, busy {dur / 1e3:.3f} ms, gaps {gaps / 1e3:.3f} ms, span {(cyc[-1][1] - ev[a][1]) / 1e6:.3f} ms")
    tot_k += dur; tot_g += gaps
n = len(best) - 1
print(f"mean per cycle: busy {tot_k / n / 1e3:.3f} ms + gaps {tot_g / n / 1e3:.3f} ms")
print(f"{'kernel':44s} {'grid':>9s} {'calls/cyc':>9s} {'avg us':>8s} {'gap before us':>13s} {'busy ms/cyc':>11s} {'gap ms/cyc':>10s}")
for (k, g), (c, t, gp) in sorted(byk.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:44s} {g:9d} {c / n:9.1f} {t / c:8.2f} {gp / c:13.2f} {t / n / 1e3:11.3f} {gp / n / 1e3:10.3f}")
