#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6/s13_driver_cmd_${1:-x}.json
python - <<PY
import json
d=json.load(open('gpurun_out/r6/s13_driver_cmd_${1:-x}.json'))
print('driver cmd: value %.0f cold %.0f computing %.0f frac %.4f bound_frac %.3f' % (d['value'], d['cold_images_per_sec'], d['folding']['images_per_sec_computing_them'], d['roofline']['frac'], d['roofline']['bound_frac']))
for o in d['other_configs']: print('   ', o.get('config'), o.get('value'), o.get('median_ms'), o.get('error'))
PY
ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('round-5 library, driver cmd: value %.0f cold %.0f computing %s' % (d['value'], d['cold_images_per_sec'], d['folding']['images_per_sec_computing_them']))"
date
} > gpurun_out/r6/s13_${1:-x}.log 2>&1
cat gpurun_out/r6/s13_${1:-x}.log | cut -c1-250
