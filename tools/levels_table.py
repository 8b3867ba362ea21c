"""Per-(level, direction) time of the sweep launches: matches the EMG3D_LOG launch log with the rocprofv3
kernel trace written by tools/prof_levels.sh (both cycles + the isolated roofline sweeps of bench.py)."""
import csv, collections, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_levels'
rows = list(csv.DictReader(open(glob.glob(d + '/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sw = [((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][5:26]) for r in rows if 'k_line_sweep' in r['Kernel_Name']]
log = [l.split() for l in open(d + '.err') if l.startswith('[sweep]')]
assert len(sw) == len(log), (len(sw), len(log))
agg = collections.OrderedDict()
for l, (t, name) in zip(log, sw):
    k = (int(l[2]), int(l[3]), int(l[4]), int(l[6]), name)
    a = agg.setdefault(k, [0, 0.0, 0])
    a[0] += 1; a[1] += t; a[2] = int(l[8])
tot = sum(a[1] for a in agg.values())
print('sweeps total us', round(tot))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    nL = k[k[3]]
    print(k[:4], k[4], 'nL', nL, 'launches', a[0], 'lines~', a[2], f"avg {a[1]/a[0]:.1f}us total {a[1]:.0f}us {100*a[1]/tot:.1f}%")
# all kernels of the trace by name (sweeps included), to see what the non-sweep launches cost
byname = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].split('(')[0][:60]
    a = byname.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
allt = sum(a[1] for a in byname.values())
print('all kernels total us', round(allt))
for n, a in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:60s} launches {a[0]:5d} avg {a[1]/a[0]:8.1f}us total {a[1]:9.0f}us {100*a[1]/allt:5.1f}%")
