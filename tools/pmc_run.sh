#!/bin/bash
# GPU box: two rocprofv3 --pmc passes (SQ wait/active, then instruction mix / LDS / MFMA) over a short bench run,
# summaries into gpurun_out/<tag>_sq{1,2}.csv + printed per-kernel table.   usage: tools/pmc_run.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in 1 2; do
  if [ $p = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS";
  else C="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; fi
  rm -rf /tmp/pmc$p
  rocprofv3 --pmc $C -d /tmp/pmc$p -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-other-configs --profile-steps 1 > /tmp/pmc$p.log 2>&1
  f=$(ls /tmp/pmc$p/*/*counter_collection.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls /tmp/pmc$p/*counter_collection.csv | head -1)
  mkdir -p /tmp/pmcd$p/x; cp $f /tmp/pmcd$p/x/pmc_counter_collection.csv
  grep -E "stage|head|tail|conv16|backend|Counter_Name" $f > $R/gpurun_out/${tag}_sq$p.csv
done
python3 $R/tools/pmc_summary.py /tmp/pmcd1 /tmp/pmcd2
