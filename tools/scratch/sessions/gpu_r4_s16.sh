#!/bin/bash
# exact stage 0: parity and speed
mkdir -p gpurun_out/r4
python tools/scratch/dbg_logit_err.py > gpurun_out/r4/dbg_logit2.txt 2>&1
grep -E "worst|s0.bn|s1.bn|s3.bn2|s8.bn|d3.relu" gpurun_out/r4/dbg_logit2.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/s16_pytest.txt; tail -5 gpurun_out/r4/s16_pytest.txt
for i in 1 2; do python bench.py --steps 400 --warmup 20 --no-cold-pass 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['path']['launch_ms'])"; done
