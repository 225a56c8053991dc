#!/usr/bin/env python3
"""Audit a hipcc -S listing: flag instructions that read (or overwrite) the destination VGPRs of an LDS read
that no s_waitcnt lgkmcnt has retired yet (straight-line approximation: labels/branches are ignored,
scalar memory ops are not counted).  usage: asm_lds_audit.py file.s <kernel-substring>"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]

def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out

for m in re.finditer(r'\.type\s+(\S+),@function\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    if pat not in m.group(1): continue
    ins = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    pend = []   # outstanding lgkm ops in issue order: (index, text, destregs)
    nflag = 0
    for i, l in enumerate(ins):
        op = l.split()[0]
        if op == 's_waitcnt':
            mm = re.search(r'lgkmcnt\((\d+)\)', l)
            if mm:
                n = int(mm.group(1))
                pend = pend[len(pend) - n:] if n < len(pend) else pend
                if n == 0: pend = []
            continue
        ops = l[len(op):]
        parts = [p.strip() for p in ops.split(',')]
        if op.startswith('ds_read') or op.startswith('ds_load'):
            dest = regs(parts[0]); srcs = set().union(*[regs(p) for p in parts[1:]]) if len(parts) > 1 else set()
        elif op.startswith('ds_write') or op.startswith('ds_store'):
            dest = set(); srcs = regs(ops)
        elif op.startswith(('s_load', 's_buffer_load', 's_memtime')):
            pend.append((i, l, set())); continue
        else:
            dest = regs(parts[0]) if op.startswith('v_') or op.startswith(('global_load', 'buffer_load')) else set()
            srcs = set().union(*[regs(p) for p in parts[1:]]) if len(parts) > 1 else set()
            if op.startswith(('global_store', 'buffer_store', 'global_load_lds')): srcs = regs(ops); dest = set()
        for (j, t, d) in pend:
            hit_r = d & srcs; hit_w = d & dest
            if hit_r or hit_w:
                nflag += 1
                if nflag <= 40:
                    print('line %d: %s\n    touches v%s of un-retired  [%d] %s   (%d LDS ops outstanding)' % (
                        i, l, sorted(hit_r | hit_w), j, t, len(pend)))
        if op.startswith('ds_'):
            pend.append((i, l, dest))
    print(m.group(1)[-50:], ': %d instructions, %d flagged' % (len(ins), nflag))
