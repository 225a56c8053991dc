#!/bin/bash
# GPU box: LDS conflict share + timing of the pair after a layout change
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pq && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pq -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cold-pass --no-cpu-baseline --no-unfolded-arm > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'stage' in k or 'backend' in k: acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    m = lambda c: sum(v[c]) / max(len(v[c]), 1)
    print("%-62s lds_conflict %4.1f %%" % (k, 100 * m('SQ_LDS_BANK_CONFLICT') / max(m('SQ_LDS_IDX_ACTIVE'), 1)))
PY
cd $R && bash tools/gpu_frozen.sh 2>&1 | tail -6
