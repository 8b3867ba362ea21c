"""Instruction mix per basic block of one kernel (dev aid): python tools/isa_stats.py <mangled-name-substring>"""
import re, collections, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = '/tmp/emg3d_isa.s'
if '--reuse' not in sys.argv:
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                    '-Wno-unused-value', '-o', out, os.path.join(root, 'emg3d_amd/csrc/emg3d_hip.hip')],
                   check=True, stderr=subprocess.DEVNULL, cwd='/tmp')
s = open(out).read()
name = sys.argv[1]
m = re.search(r'^(\S*' + re.escape(name) + r'\S*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
cur = 'entry'; blocks = collections.OrderedDict(entry=[])
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = l.split(':')[0] + (' LOOP' if 'Loop Header' in l else ''); blocks[cur] = []
    elif l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.'):
        blocks[cur].append(l.strip())
tot = 0
for k, v in blocks.items():
    if not v: continue
    c = collections.Counter(x.split()[0] for x in v)
    tot += len(v)
    print(k, len(v), c.most_common(9))
print('total', tot)
for key in ('vgpr_count', 'agpr_count', 'vgpr_spill', 'lds_size', 'scratch'):
    mm = re.search(r'\.' + key + r':\s*(\d+)', s[j:j + 6000])
print(re.findall(r'; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)', s[j:j + 3000]))
