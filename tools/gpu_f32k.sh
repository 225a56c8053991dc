#!/bin/bash
# GPU box: float32 handles after the input-channel fold of stage 3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_f32.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
python bench.py --dtype f32 --steps 60 --warmup 10 --no-cpu-baseline --no-cold-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('f32 %.0f img/s frac %.3f  computing them %s  launches ' % (d['value'], d['roofline']['frac'], d.get('folding',{}).get('images_per_sec_computing_them')) + ' '.join('%.3f'%l['ms'] for l in d['roofline']['launches']))"
done
