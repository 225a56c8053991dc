#!/bin/bash
# tail_kernel duration (kernel trace, batch 64: banded launches + tail launch), round-3 library vs current
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in r3 base; do
  if [ "$v" = base ]; then unset ROOMNET_HIP_LIB; else export ROOMNET_HIP_LIB=$R/roomnet_amd/lib/libroomnet_hip_$v.so; fi
  rm -rf /tmp/kt_$v && rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt --output-format csv -- python3 $R/bench.py --batch 64 --steps 300 --warmup 10 --no-cpu-baseline --no-parity-check --no-cold-pass > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "tail|backend|conv16|stage6x" $f | cut -d, -f1-4,6,7 | cut -c1-170
done
