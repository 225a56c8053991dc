#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the fused stage pair for library variants: tools/gpu_fetch.sh name1 name2 ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=$R/tools/ab/libroomnet_hip_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pf_$c && rocprofv3 --pmc $c -d /tmp/pf_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-parity-check --profile-steps 1 > /tmp/pf.log 2>&1
    f=$(find /tmp/pf_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" "$c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "stage23pc" in r["Kernel_Name"]]
vals = [float(r["Counter_Value"]) for r in rows]
print("%-8s %-10s fused pair: mean %.1f MiB per dispatch over %d dispatches" % (sys.argv[2], sys.argv[3], sum(vals) / len(vals) / 1024, len(vals)))
PY
  done
done
