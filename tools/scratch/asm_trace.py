#!/usr/bin/env python3
"""Condensed instruction trace between two s_barriers of one kernel in a hipcc -S listing.
usage: asm_trace.py file.s <kernel-substring> <barrier-index>"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for m in re.finditer(r'\.type\s+(\S+),@function\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    if pat not in m.group(1):
        continue
    ins = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    idx = [i for i, l in enumerate(ins) if l.startswith('s_barrier')]
    seg = ins[idx[k]:idx[k + 1] + 1]
    out = []
    for l in seg:
        op = l.split()[0]
        if op.startswith('s_waitcnt'): out.append('W:' + l.split(None, 1)[1].replace(' ', ''))
        elif op.startswith('v_mfma'): out.append('MFMA')
        elif op.startswith('ds_read'): out.append('dsr')
        elif op.startswith('ds_write'): out.append('dsw')
        elif op.startswith('global_load'): out.append('GLD')
        elif op.startswith('global_store'): out.append('GST')
        elif op.startswith('s_load'): out.append('SLD')
        elif op.startswith(('s_cbranch', 's_branch')): out.append('BR')
        elif op.endswith(':'): out.append('L')
        elif op.startswith('v_'): out.append('v')
        elif op.startswith('s_'): out.append('s')
        else: out.append(op)
    res, prev, cnt = [], None, 0
    for o in out + [None]:
        if o == prev: cnt += 1
        else:
            if prev is not None: res.append(prev + (('x%d' % cnt) if cnt > 1 else ''))
            prev, cnt = o, 1
    print(m.group(1)[-60:], len(seg), 'instructions')
    print(' '.join(res))
    break
