#!/bin/bash
# GPU box: sample engine clock / power while the bench loop runs (DVFS: what clock does the chip hold in these kernels?)
cd $GRAFT_REPO_ROOT
python bench.py --steps ${1:-15000} --warmup 20 --no-cpu-baseline --no-parity-check > /tmp/bench_probe.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W): /W /' | tr '\n' ' '; echo
  sleep 1
done | sort | uniq -c | sort -k1,1nr | head -12
tail -1 /tmp/bench_probe.log | cut -c1-200
