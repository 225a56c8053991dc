#!/bin/bash
# the kernel trace of the default bench command without the other configurations (their batch-1 latency calls run the same kernels
# and pull the per-kernel averages down)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r6_b_kernel_stats.csv
tail -1 /tmp/kt.log > $R/gpurun_out/r6_b_kernel_stats_profiled_line.json
head -8 $R/gpurun_out/r6_b_kernel_stats.csv | cut -c1-160
