#!/bin/bash
# GPU box: the per-round evidence set.  usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
#   1. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)         -> <tag>_pmc_fetch_size.csv, <tag>_pmc_write_size.csv
#      -> tools/hbm_traffic.py -> <tag>_hbm_traffic.json (also dropped into profiles/ of the box's snapshot, so that the
#      bench line of step 2 reports its `roofline.traffic` from THIS run's counters)
#   2. python bench.py (default run, with cpu_baseline)                       -> <tag>_bench.json
#   3. rocprofv3 --kernel-trace --stats of the same command                   -> <tag>_kernel_stats.csv
#   4. SQ wait/active + instruction-mix passes (tools/pmc_run.sh)              -> <tag>_sq1.csv, <tag>_sq2.csv, <tag>_sq_summary.txt
#   5. in-kernel stamps (diagnostic build, if present)                        -> <tag>_stamps.txt
tag=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c && rocprofv3 --pmc $c -d /tmp/p_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-other-configs --profile-steps 1 > /tmp/p_$c.log 2>&1
  f=$(find /tmp/p_$c -name "*counter_collection.csv" | head -1)
  lc=$(echo $c | tr A-Z a-z)
  grep -E "stage|head|tail|conv16|backend|Counter_Name" "$f" > $O/${tag}_pmc_${lc}.csv
done
python3 $R/tools/hbm_traffic.py $O/${tag}_pmc_fetch_size.csv $O/${tag}_pmc_write_size.csv > $O/${tag}_hbm_traffic.json
cp $O/${tag}_hbm_traffic.json $R/profiles/${tag}_hbm_traffic.json
cd $R && python3 bench.py 2>/dev/null | tail -1 > $O/${tag}_bench.json
cd /tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/${tag}_kernel_stats.csv
$R/tools/pmc_run.sh $tag > $O/${tag}_sq_summary.txt 2>&1
if [ -f $R/tools/ab/libroomnet_hip_stamps.so ]; then
  cd $R && ROOMNET_HIP_LIB=tools/ab/libroomnet_hip_stamps.so python3 bench.py --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-other-configs --profile-steps 1 2>&1 | grep "stamps" > $O/${tag}_stamps.txt
fi
head -c 600 $O/${tag}_bench.json; echo; head -14 $O/${tag}_kernel_stats.csv | cut -c1-160
