#!/bin/bash
# GPU box, round 4 session 8: whole GPU suite, A/B against round 3, float32 bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4/s8_pytest.txt
tail -5 gpurun_out/r4/s8_pytest.txt
tools/gpu_var.sh r3 base r3 base 2>&1 | tee gpurun_out/r4/s8_ab.txt
for rep in 1 2; do
python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r4/s8_f32_bench_$rep.json
python -c "
import json
d=json.load(open('gpurun_out/r4/s8_f32_bench_$rep.json'))
print('f32 %.0f img/s cold %.0f  %.2f ms/step  ' % (d['value'], d.get('cold_images_per_sec',0), d['ms_per_step']) + ' '.join('%.3f'%x for x in d['path']['stage_ms']) + ' head %.3f' % d['path']['head_ms'], d['parity'].get('max_abs_dprob'), d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/r4/s8_f32.txt
