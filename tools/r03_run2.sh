#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run2}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest.log | tail -20; exit 1; }
python tools/ab_env.py AFD_FOLD_EVERY 1 2 4 8 100 > $O/ab_fold.txt 2>&1; cat $O/ab_fold.txt | tail -6
