#!/bin/bash
# GPU box, round 6: whole GPU suite + the evidence sets of the final code -> gpurun_out/ (copied into profiles/ afterwards):
#   r6_a_* (224 bf16: bench line, driver-command line, kernel stats, PMC traffic, SQ counters), r6_a_600_*, r6_a_f32_*, f16 / pcie /
#   no-dither lines, same-box A/B against the round-5 library, ladder, imageops (kernel trace + PMC), power, parity report
tag=${1:-r6_a}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
RN_PARITY_REPORT=gpurun_out/${tag}_parity.json python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r6/final_pytest.txt
tail -3 gpurun_out/r6/final_pytest.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_driver_cmd.json
bash tools/profile_round.sh ${tag} > gpurun_out/r6/final_profile_round.txt 2>&1
tail -12 gpurun_out/r6/final_profile_round.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
# the driver's exact command under the kernel trace
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktd && rocprofv3 --kernel-trace --stats -d /tmp/ktd -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > /tmp/ktd.log 2>/dev/null; f=$(find /tmp/ktd -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_driver_cmd_kernel_stats.csv; tail -1 /tmp/ktd.log > $GRAFT_REPO_ROOT/gpurun_out/${tag}_driver_cmd_profiled_line.json )
bash tools/profile_600.sh ${tag}_600 > gpurun_out/r6/final_profile_600.txt 2>&1
tail -6 gpurun_out/r6/final_profile_600.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/profile_f32.sh ${tag}_f32 > gpurun_out/r6/final_profile_f32.txt 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/${tag}_f32_line.json gpurun_out/${tag}_bench_f32.json
python bench.py --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_f16.json
python bench.py --no-dither --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_no_dither.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs --pcie 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_pcie.json
python -c "
import json
for f in ('gpurun_out/${tag}_bench_f32.json','gpurun_out/${tag}_bench_f16.json','gpurun_out/${tag}_bench_no_dither.json','gpurun_out/${tag}_bench.json','gpurun_out/${tag}_bench_driver_cmd.json','gpurun_out/${tag}_bench_pcie.json'):
    d=json.load(open(f)); print(f, '%.0f img/s cold %.0f unfolded %s' % (d['value'], d.get('cold_images_per_sec',0), (d.get('folding') or {}).get('images_per_sec_computing_them')), ' '.join('%.3f'%x for x in d['path']['launch_ms']), round(d['roofline']['frac'],4), d['roofline'].get('bound_frac'), {k: round(v) for k, v in d['path'].items() if k.startswith('pcie')})
    for o in d.get('other_configs', []): print('     other:', o.get('config'), o.get('value'), o.get('median_ms'), (o.get('roofline') or {}).get('frac'), o.get('images_per_sec_computing_them'))"
{
echo "# tools/gpu_r6_final.sh on one MI355X box, same session: bench.py --steps 200 --warmup 10 (no cold pass, no other configs), the product library against the round-5 library (tools/ab/libroomnet_hip_r5.so), alternated"
line() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-cold-pass --no-parity-check --no-other-configs --no-unfolded-arm "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-22s %.0f img/s  launches ' % ('$V', d['value']) + ' '.join('%.3f'%x for x in d['path']['launch_ms']))"; }
for rep in 1 2 3; do
V="round 6"; line
V="round 6 --no-dither"; line --no-dither
V="round 6 --compute-frozen"; line --compute-frozen
[ -f tools/ab/libroomnet_hip_r5.so ] && { V="round-5 library"; ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_r5.so line; }
done
} 2>&1 | tee gpurun_out/${tag}_ab_r5.txt
{
echo "# tools/gpu_r6_final.sh on one MI355X box: bench.py --steps K --warmup 5 --dtype D (batch 256, 224 x 224, one handle, no spin-up), fresh process per line"
echo "# dtype steps   value img/s   ms/step   cold img/s   path/HBM-roofline(value)"
for dt in bf16 f16; do for k in 20 200 6000; do
  python bench.py --steps $k --warmup 5 --dtype $dt --no-cpu-baseline --no-other-configs --profile-steps 1 --event-steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-5s %5d   %9.0f   %.4f   %9.0f   %.4f' % ('$dt', $k, d['value'], d['ms_per_step'], d.get('cold_images_per_sec', 0), d['path']['hbm_frac']))"
done; done
} 2>&1 | tee gpurun_out/${tag}_ladder.txt
# ---- the batched crop + resize kernel (f1): timing line, kernel trace, PMC traffic
python tools/profile_imageops.py > gpurun_out/${tag}_imageops.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kti && rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_imageops.py > /tmp/kti.log 2>&1; f=$(find /tmp/kti -name "*kernel_stats.csv" | head -1); grep -E "Name|resize" "$f" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_imageops_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pi_$c && rocprofv3 --pmc $c -d /tmp/pi_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_imageops.py --repeat 2 > /tmp/pi_$c.log 2>&1; f=$(find /tmp/pi_$c -name "*counter_collection.csv" | head -1); lc=$(echo $c | tr A-Z a-z); grep -E "Counter_Name|resize_batch" "$f" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_imageops_pmc_${lc}.csv; done )
cat gpurun_out/${tag}_imageops.json | cut -c1-400
bash tools/power_pass.sh > gpurun_out/${tag}_power.txt 2>&1
grep "img/s" gpurun_out/${tag}_power.txt
{
echo "# tools/bench_images.py --dir on the GPU box ($(nproc) usable host threads): JPEG files -> classify_im_dir / groundtruth_validation"
for t in 16; do
python tools/bench_images.py --dir --threads=$t 1080 1920 256 2>&1 | tail -1
python tools/bench_images.py --dir --threads=$t 480 640 512 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/${tag}_bench_images.txt
