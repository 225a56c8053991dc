#!/bin/bash
# GPU box: evidence for the float32 path (RN_DTYPE_F32 handles, matrix-core conv stages): tools/profile_f32.sh <tag>
#   bench line, kernel trace of the same command, one PMC pass (matrix-pipe busy cycles per launch)
tag=${1:-rX_f32}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
ARGS="--dtype f32 --steps 20 --warmup 3"
cd $R && python3 bench.py $ARGS 2>/dev/null | tail -1 > $O/${tag}_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktf && rocprofv3 --kernel-trace --stats -d /tmp/ktf -o kt --output-format csv -- python3 $R/bench.py $ARGS --no-cpu-baseline > /tmp/ktf.log 2>&1
f=$(find /tmp/ktf -name "*kernel_stats.csv" | head -1); cp "$f" $O/${tag}_kernel_trace.csv
rm -rf /tmp/pf && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pf -o p --output-format csv -- python3 $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cold-pass --no-cpu-baseline > /tmp/pf.log 2>&1
f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
python3 - "$f" > $O/${tag}_mfma_busy.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'f32' in k or 'head' in k:
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print("# per launch (mean over dispatches): matrix-pipe busy cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs), LDS bank-conflict share")
for k, v in acc.items():
    m = lambda c: sum(v[c]) / max(len(v[c]), 1) if c in v else 0.0
    print("%-110s n=%3d  mfma_busy %8.0f K cycles/SIMD  lds_conflict %4.1f %%" % (k[:110], len(v.get('SQ_BUSY_CYCLES', [])), m('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / 1e3,
                                                                                       100 * m('SQ_LDS_BANK_CONFLICT') / max(m('SQ_LDS_IDX_ACTIVE'), 1)))
PY
head -c 300 $O/${tag}_line.json; echo; head -14 $O/${tag}_kernel_trace.csv | cut -c1-170; cat $O/${tag}_mfma_busy.txt | cut -c1-200
