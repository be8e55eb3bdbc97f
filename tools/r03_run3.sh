#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run3}
mkdir -p $O
cd $R
python tools/ab_env.py AFD_FOLD_EVERY 0 1 2 3 > $O/ab_fold.txt 2>&1; tail -5 $O/ab_fold.txt
python tools/ab_env.py AFD_WGRAD_BATCH 2 4 6 8 > $O/ab_batch.txt 2>&1; tail -5 $O/ab_batch.txt
