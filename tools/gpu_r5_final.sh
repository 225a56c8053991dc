#!/bin/bash
# GPU box, round 5: whole GPU suite + the evidence sets of the final code (r5_d: 224 bf16; r5_d_600; r5_d_f32) + f16 / driver-command
# lines + a short ladder + the directory drivers on real files + the PCIe-inclusive arms
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r5/final_pytest.txt
tail -3 gpurun_out/r5/final_pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r5/parity_final.json 2>/dev/null
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5_d_bench_driver_cmd.json
bash tools/profile_round.sh r5_d > gpurun_out/r5/final_profile_round.txt 2>&1
tail -12 gpurun_out/r5/final_profile_round.txt | cut -c1-200
bash tools/profile_600.sh r5_d_600 > gpurun_out/r5/final_profile_600.txt 2>&1
tail -12 gpurun_out/r5/final_profile_600.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/profile_f32.sh r5_d_f32 > gpurun_out/r5/final_profile_f32.txt 2>&1
tail -30 gpurun_out/r5/final_profile_f32.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
cp gpurun_out/r5_d_f32_line.json gpurun_out/r5_d_bench_f32.json
python bench.py --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_d_bench_f16.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --pcie 2>/dev/null | tail -1 > gpurun_out/r5_d_bench_pcie.json
python -c "
import json
for f in ('gpurun_out/r5_d_bench_f32.json','gpurun_out/r5_d_bench_f16.json','gpurun_out/r5_d_bench.json','gpurun_out/r5_d_bench_driver_cmd.json','gpurun_out/r5_d_bench_pcie.json'):
    d=json.load(open(f)); print(f, '%.0f img/s cold %.0f' % (d['value'], d.get('cold_images_per_sec',0)), ' '.join('%.3f'%x for x in d['path']['stage_ms']), d['roofline']['frac'], {k: round(v) for k, v in d['path'].items() if k.startswith('pcie')})"
{
echo "# tools/gpu_r5_final.sh on one MI355X box: bench.py --steps K --warmup 5 --dtype D (batch 256, 224 x 224, one handle, no spin-up), fresh process per line"
echo "# dtype steps   value img/s   ms/step   cold img/s   path/HBM-roofline(value)"
for dt in bf16 f16; do for k in 20 200 6000; do
  python bench.py --steps $k --warmup 5 --dtype $dt --no-cpu-baseline --profile-steps 1 --event-steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-5s %5d   %9.0f   %.4f   %9.0f   %.4f' % ('$dt', $k, d['value'], d['ms_per_step'], d.get('cold_images_per_sec', 0), d['path']['hbm_frac']))"
done; done
} 2>&1 | tee gpurun_out/r5/ladder.txt
{
echo "# tools/bench_images.py --dir on the GPU box ($(nproc) usable host threads): JPEG files -> classify_im_dir / groundtruth_validation"
for t in 16 32; do
python tools/bench_images.py --dir --threads=$t 1080 1920 256 2>&1 | tail -1
python tools/bench_images.py --dir --threads=$t 480 640 512 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/r5/bench_images.txt
