#!/bin/bash
# round 6, session 7: dithered stores + refined weight rounding: GPU suite, parity numbers, timing against the round-5 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
RN_PARITY_REPORT=gpurun_out/r6/s7_parity.json timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6/s7_parity.json'))
for sec in d:
    if sec in ('what',): continue
    print(sec, json.dumps(d[sec])[:1500])
PY
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new (dither)"; line
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
date
} > gpurun_out/r6/s7.log 2>&1
tail -70 gpurun_out/r6/s7.log
