#!/usr/bin/env python3
"""isa_blocks.py FILE.s KERNEL_SUBSTRING [MIN] — per basic block of one compiled kernel: VALU / SALU / LDS / VMEM
instruction counts (blocks with at least MIN instructions), to see where the instructions of a traversal step go."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 12
src = open(path).read().split('\n')
start = [i for i, l in enumerate(src) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':')][0]
end = [i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end')][0]
blocks = []; cur = ['entry', start, collections.Counter(), collections.Counter()]; blocks.append(cur)
for i in range(start + 1, end):
    l = src[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), i, collections.Counter(), collections.Counter()]; blocks.append(cur); continue
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') : continue
    op = t.split()[0]
    kind = 'v' if op.startswith('v_') else 's' if op.startswith('s_') else 'ds' if op.startswith('ds_') else 'vm' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'o'
    cur[2][kind] += 1; cur[3][op] += 1
for b in blocks:
    n = sum(b[2].values())
    if n >= lo:
        print(f"{b[0]:12s} line {b[1]+1:6d}  v {b[2]['v']:4d}  s {b[2]['s']:4d}  ds {b[2]['ds']:3d}  vm {b[2]['vm']:3d}   top: " + ' '.join(f"{o}:{c}" for o, c in b[3].most_common(8)))
