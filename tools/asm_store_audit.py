#!/usr/bin/env python3
"""Flag VALU / LDS-return writes to the data VGPRs of a wide store issued fewer than W instructions earlier.
A VALU write right behind the store (+1) is the hazard hipcc does not pad for buffer stores with an SGPR soffset
(NOTES.md, rounds 1-2 section 4); an LDS read returning into the registers is harmless (its data arrives 64+ cycles later) and is
listed as "lds".   usage: asm_store_audit.py file.s <kernel-substring> [W=4]"""
import re, sys
s = open(sys.argv[1]).read(); pat = sys.argv[2]; W = int(sys.argv[3]) if len(sys.argv) > 3 else 4
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
for m in re.finditer(r'\.type\s+(\S+),@function\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    if pat not in m.group(1): continue
    ins = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    n = 0
    for i, l in enumerate(ins):
        op = l.split()[0]
        if op.startswith(('buffer_store_dwordx4', 'buffer_store_dwordx3', 'global_store_dwordx4', 'global_store_dwordx3', 'ds_write_b128', 'ds_write_b96')):
            parts = [p.strip() for p in l[len(op):].split(',')]
            data = regs(parts[0]) if op.startswith('buffer') else regs(parts[1])
            for j in range(i + 1, min(i + 1 + W, len(ins))):
                o2 = ins[j].split()[0]
                if o2.startswith('v_') or o2.startswith('ds_read'):
                    d2 = regs(ins[j][len(o2):].split(',')[0])
                    if d2 & data:
                        kind = 'lds' if o2.startswith('ds_read') else 'VALU'
                        if kind == 'VALU': n += 1
                        print('[%d] %s\n   +%d  %s  (%s)' % (i, l, j - i, ins[j], kind))
    print(m.group(1)[-50:], n, 'VALU overwrites flagged')
