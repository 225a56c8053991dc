#!/bin/bash
# GPU box, round 4: whole GPU suite + the evidence sets of the final code (r4_c: 224 bf16; r4_600) + f16 / f32 bench lines + ladder
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r4/final_pytest.txt
tail -3 gpurun_out/r4/final_pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r4/parity_final.json 2>/dev/null
bash tools/profile_round.sh r4_c > gpurun_out/r4/final_profile_round.txt 2>&1
tail -12 gpurun_out/r4/final_profile_round.txt | cut -c1-200
bash tools/profile_600.sh r4_600 > gpurun_out/r4/final_profile_600.txt 2>&1
tail -12 gpurun_out/r4/final_profile_600.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/profile_f32.sh r4_f32 > gpurun_out/r4/final_profile_f32.txt 2>&1
tail -30 gpurun_out/r4/final_profile_f32.txt | cut -c1-200
cp gpurun_out/r4_f32_line.json gpurun_out/r4_c_bench_f32.json
python bench.py --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_c_bench_f16.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_c_bench_driver_cmd.json
python -c "
import json
for f in ('gpurun_out/r4_c_bench_f32.json','gpurun_out/r4_c_bench_f16.json','gpurun_out/r4_c_bench.json','gpurun_out/r4_c_bench_driver_cmd.json'):
    d=json.load(open(f)); print(f, '%.0f img/s cold %.0f' % (d['value'], d.get('cold_images_per_sec',0)), ' '.join('%.3f'%x for x in d['path']['stage_ms']), d['roofline']['frac'])"
bash tools/gpu_ladder.sh > /dev/null 2>&1
cat gpurun_out/r4/ladder.txt
