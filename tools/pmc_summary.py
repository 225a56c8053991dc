#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc counter_collection.csv passes (SQ wait/active + LDS/MFMA) per kernel."""
import csv, glob, collections, sys
res = collections.defaultdict(dict)
for d in sys.argv[1:]:
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'stage' not in name and 'head' not in name and 'tail' not in name and 'conv16' not in name and 'backend' not in name: continue
        key = name[max(name.find('stage'), name.find('tail_kernel'), name.find('conv16_kernel'), name.find('backend_kernel')):name.find('>') + 1] if '<' in name else name[:30]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        res[key]['vgpr'] = r['VGPR_Count']; res[key]['lds'] = r['LDS_Block_Size']; res[key]['wg'] = r['Workgroup_Size']; res[key]['grid'] = r['Grid_Size']
    for k, v in agg.items():
        for c, x in v.items(): res[k][c] = sum(x) / len(x)
for k, v in res.items():
    wc = v.get('SQ_WAVE_CYCLES', 1)
    nm = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 32
    print('%-48s wg %s vgpr %s lds %s' % (k, v.get('wg'), v.get('vgpr'), v.get('lds')))
    print('    wait_any %2.0f%%  wait_inst %2.0f%%  active %2.0f%%  wait_lds %4.1f%%  valu/mfma %5.1f  mfma_busy %4.0fK cyc/SIMD (%.3f ms @2.4GHz)  lds_conflict %4.1f%%' % (
        100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        100 * v.get('SQ_WAIT_INST_LDS', 0) / wc, v.get('SQ_INSTS_VALU', 0) / max(nm, 1), nm * 32 / 1024 / 1e3, nm * 32 / 1024 / 2.4e6,
        100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
