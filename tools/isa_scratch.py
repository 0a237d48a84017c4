#!/usr/bin/env python3
"""isa_scratch.py FILE.s KERNEL_SUBSTRING — scratch (spill) loads / stores of one compiled kernel by basic block and loop depth."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(src) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':')][0]
end = [i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end')][0]
cur = 'entry'; info = {}
for i in range(start, end):
    l = src[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1); d = 0
        for j in range(i, min(i + 4, end)):
            mm = re.search(r'Depth=(\d+)', src[j])
            if mm: d = max(d, int(mm.group(1)))
        info[cur] = [i, d, 0, 0]
        continue
    t = l.split(';')[0].strip()
    if t.startswith('scratch_'):
        info.setdefault(cur, [i, 0, 0, 0])[2 if 'load' in t else 3] += 1
tot = [0, 0]
for k, v in info.items():
    if v[2] + v[3]:
        print(k, 'line', v[0] + 1, 'loop depth', v[1], 'loads', v[2], 'stores', v[3]); tot[0] += v[2]; tot[1] += v[3]
print('total', tot)
