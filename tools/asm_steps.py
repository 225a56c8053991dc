#!/usr/bin/env python3
"""Instruction-class counts between consecutive s_barrier instructions of one kernel in a
hipcc -S --cuda-device-only listing: tools/asm_steps.py FILE.s KERNEL_SUBSTRING
(one line per barrier-delimited segment = one row step of an unrolled row loop)."""
import re, sys
from collections import Counter

def classify(l):
    if l.startswith('v_mfma'): return 'mfma'
    if l.startswith('v_'): return 'valu'
    if l.startswith('ds_read'): return 'dsr'
    if l.startswith('ds_write'): return 'dsw'
    if l.startswith(('global_load', 'global_store', 'buffer_')): return 'vmem'
    if l.startswith('s_waitcnt'): return 'wait'
    if l.startswith('s_nop'): return 'nop'
    if l.startswith('s_'): return 'salu'
    return 'other'

s = open(sys.argv[1]).read()
for m in re.finditer(r'\.type\s+(\S+),@function\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    if sys.argv[2] not in m.group(1):
        continue
    print(m.group(1)[:100])
    ins = [l.strip() for l in m.group(2).split('\n')
           if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    seg, cur = [], []
    for l in ins:
        cur.append(l)
        if l.startswith('s_barrier'):
            seg.append(cur); cur = []
    seg.append(cur)
    for i, sg in enumerate(seg):
        c = Counter(classify(l) for l in sg)
        print('%3d total %4d  ' % (i, len(sg)) + ' '.join('%s %d' % kv for kv in sorted(c.items())))
