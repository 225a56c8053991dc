#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python tools/scratch/dbg_backend.py 2>&1 | tail -12 | tee gpurun_out/r4/s12_dbg.txt
