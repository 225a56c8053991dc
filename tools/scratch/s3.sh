#!/bin/bash
# round 6, session 3: whole GPU suite on the const4 + AB-library tree, then same-box A/B against the round-5 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
date
line() { python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="new"; line
V="r5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line
done
echo "== driver cmd"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6/s3_driver_cmd.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6/s3_driver_cmd.json'))
print('value', d['value'], 'cold', d.get('cold_images_per_sec'), 'unfolded', d['folding']['images_per_sec_computing_them'], 'bound_frac', d['roofline'].get('bound_frac'))
for o in d.get('other_configs', []):
    print('  other', o.get('config'), o.get('value'), o.get('median_ms'), o.get('roofline',{}).get('frac'), o.get('images_per_sec_computing_them'))
PY
date
} > gpurun_out/r6/s3.log 2>&1
tail -40 gpurun_out/r6/s3.log
