#!/bin/bash
# GPU box: evidence for BASELINE config 5 at its per-GPU size (64 x 600x600, fp16): tools/profile_600.sh <tag>
#   traffic (PMC) first, then the bench line, then the kernel trace -- like tools/profile_round.sh
tag=${1:-rX_600}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
ARGS="--side 600 --batch 64 --dtype f16"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c && rocprofv3 --pmc $c -d /tmp/p_$c -o p --output-format csv -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --profile-steps 1 > /tmp/p_$c.log 2>&1
  f=$(find /tmp/p_$c -name "*counter_collection.csv" | head -1)
  lc=$(echo $c | tr A-Z a-z)
  grep -E "stage|head|tail|conv16|backend|Counter_Name" "$f" > $O/${tag}_pmc_${lc}.csv
done
python3 $R/tools/hbm_traffic.py $O/${tag}_pmc_fetch_size.csv $O/${tag}_pmc_write_size.csv 64 600 f16 > $O/${tag}_hbm_traffic.json
cp $O/${tag}_hbm_traffic.json $R/profiles/${tag}_hbm_traffic.json
cd $R && python3 bench.py $ARGS --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/${tag}_bench.json
python3 bench.py --side 600 --batch 64 --dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${tag}_bench_bf16.json
cd /tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/${tag}_kernel_stats.csv
head -c 400 $O/${tag}_bench.json; echo; head -12 $O/${tag}_kernel_stats.csv | cut -c1-150
