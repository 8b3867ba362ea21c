"""Instruction counts of the loops of one kernel in a hipcc -S listing: python3 tools/isa_loops.py file.s mangled-name-regex [min_len]"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(%s):' % sys.argv[2], s, re.M)
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 25
body = s[m.start():s.index('.Lfunc_end', m.start())].split('\n')
ins = []
for l in body:
    t = l.strip().split(';')[0].strip()
    if not t or (t.startswith('.') and not t.startswith('.LBB')):
        continue
    ins.append(t)
labels = {t.rstrip(':').strip(): k for k, t in enumerate(ins) if t.startswith('.LBB')}
print(len(ins), "instructions")
for k, t in enumerate(ins):
    mm = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', t) or re.match(r's_branch (\.LBB\d+_\d+)', t)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        a = labels[mm.group(1)]
        seg = [x for x in ins[a:k + 1] if not x.startswith('.LBB')]
        if len(seg) < minlen:
            continue
        c = collections.Counter()
        for x in seg:
            op = x.split()[0]
            if op.startswith('v_') and 'f64' in op: c['f64'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('ds_'): c['lds'] += 1
            elif op.startswith('global_') or op.startswith('buffer_'): c['vmem'] += 1
            elif op.startswith('s_waitcnt'): c['wait'] += 1
            elif op.startswith('s_'): c['salu'] += 1
            else: c[op] += 1
        print(mm.group(1), 'at', a, 'len', len(seg), dict(c), 'sleep' if any('s_sleep' in x for x in seg) else '')
