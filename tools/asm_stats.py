#!/usr/bin/env python3
"""Per-kernel instruction-class counts from a hipcc -S --cuda-device-only listing."""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.type\s+(\S+),@function\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    name, body = m.group(1), m.group(2)
    ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    c = lambda p: sum(1 for l in ins if re.match(p, l))
    short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:48]
    print('%-48s total %5d mfma %4d valu %5d dsr %4d dsw %4d gld %3d gst %3d scratch %3d wait %4d bar %2d sld %3d br %3d' % (
        short, len(ins), c('v_mfma'), c('v_') - c('v_mfma'), c('ds_read'), c('ds_write'), c('global_load'),
        c('global_store'), c('scratch_'), c('s_waitcnt'), c('s_barrier'), c('s_load'), c('s_cbranch')))
