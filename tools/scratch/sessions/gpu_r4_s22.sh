#!/bin/bash
# SQ counters of the float32 matrix-core stages, old (epilogue behind the chain) vs new (epilogue inside the chain)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in oldf base; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=$R/roomnet_amd/lib/libroomnet_hip_$v.so; fi
  for p in 1 2; do
    if [ $p = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS";
    else C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES"; fi
    rm -rf /tmp/q$p && rocprofv3 --pmc $C -d /tmp/q$p -o q --output-format csv -- python3 $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cold-pass --no-cpu-baseline --no-parity-check > /tmp/q$p.log 2>&1
    f=$(find /tmp/q$p -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'f32m' in k:
        key = k[k.index('<'):k.index('>') + 1]
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(sys.argv[2], k, ' '.join('%s=%.3g' % (c.replace('SQ_', ''), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
  done
done
