#!/bin/bash
# GPU box, round 4 session 9: whole GPU suite + evidence sets r4_b (224, bf16) and r4_600 + float32 bench line + parity report
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4/s9_pytest.txt
tail -3 gpurun_out/r4/s9_pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r4/parity.json 2>/dev/null
bash tools/profile_round.sh r4_b > gpurun_out/r4/s9_profile_round.txt 2>&1
tail -12 gpurun_out/r4/s9_profile_round.txt | cut -c1-200
bash tools/profile_600.sh r4_600 > gpurun_out/r4/s9_profile_600.txt 2>&1
tail -12 gpurun_out/r4/s9_profile_600.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
python bench.py --dtype f32 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r4_f32_bench.json
python bench.py --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_b_bench_f16.json
python -c "
import json
for f in ('gpurun_out/r4_f32_bench.json','gpurun_out/r4_b_bench_f16.json','gpurun_out/r4_b_bench.json'):
    d=json.load(open(f)); print(f, '%.0f img/s cold %.0f' % (d['value'], d.get('cold_images_per_sec',0)), ' '.join('%.3f'%x for x in d['path']['stage_ms']), d['roofline']['frac'])"
