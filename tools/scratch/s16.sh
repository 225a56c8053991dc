#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
t0=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/final_driver_cmd.json 2> gpurun_out/r6/final_driver_cmd.err
t1=$(date +%s.%N)
echo "wall seconds of the driver's command: $(echo "$t1 - $t0" | bc)"
tail -1 gpurun_out/r6/final_driver_cmd.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('value %.0f cold %.0f computing %.0f frac %.4f bound %.3f' % (d['value'], d['cold_images_per_sec'], d['folding']['images_per_sec_computing_them'], d['roofline']['frac'], d['roofline']['bound_frac']))
print([(o.get('config'), o.get('value') or o.get('median_ms'), o.get('error')) for o in d['other_configs']])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
