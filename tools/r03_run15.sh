#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run15}
mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_env.py AFD_WGRAD_BATCH 2 4 8 12 16 --rounds 3 > $O/ab_batch.txt 2>&1; tail -5 $O/ab_batch.txt
timeout -k 10 300 python tools/ab_env.py AFD_FOLD_EVERY 1 2 4 --rounds 3 > $O/ab_fold.txt 2>&1; tail -3 $O/ab_fold.txt
AFD_WGRAD_STREAMS=2 timeout -k 10 200 python tools/ab_env.py AFD_NOOP 0 --rounds 3 2>&1 | tail -1 | sed 's/^/2 side streams: /'
timeout -k 10 200 python tools/chain_time.py 2>&1 | tail -4
