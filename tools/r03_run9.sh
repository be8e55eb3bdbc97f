#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run9}
mkdir -p $O
cd $R
timeout -k 10 200 python tools/dbg_buckets.py 2 > $O/dbg.txt 2>&1; tail -12 $O/dbg.txt
