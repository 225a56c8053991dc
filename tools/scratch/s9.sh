#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
{
date
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40
python -c "import __graft_entry__ as g; g.smoke()"
date
} > gpurun_out/r6/s9.log 2>&1
tail -50 gpurun_out/r6/s9.log | cut -c1-300
