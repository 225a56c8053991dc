#!/bin/bash
# GPU box: cache-policy variants of a launch: FETCH_SIZE / WRITE_SIZE per kernel and sustained rate.  tools/gpu_nt.sh name1 name2 ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=$R/tools/ab/libroomnet_hip_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_$c && rocprofv3 --pmc $c -d /tmp/p_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-cold-pass --profile-steps 1 > /tmp/p_$c.log 2>&1
    f=$(find /tmp/p_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" $c <<'PY'
import csv,sys,collections
f,v,c=sys.argv[1:4]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']==c:
        k=r['Kernel_Name']
        for t in ('stage23x','stage4x','stage5x','stage_rw','backend'):
            if t in k: acc[t].append(float(r['Counter_Value']))
print(v,c,' '.join('%s %.0f MB (n=%d)'%(t,sum(x)/len(x)*1024/1e6*(2 if c=='FETCH_SIZE' else 1),len(x)) for t,x in acc.items() if len(x)>2))
PY
  done
done
cd $R
for rep in 1 2; do STEPS=${STEPS:-400} tools/gpu_var.sh "$@"; done
