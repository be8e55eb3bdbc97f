#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run8}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest.log | tail -30; exit 1; }
timeout -k 10 300 python tools/gn_bench.py > $O/gn_bench.txt 2>&1; tail -8 $O/gn_bench.txt
timeout -k 10 300 python tools/ab_env.py AFD_NOOP 0 1 --rounds 3 > $O/step.txt 2>&1; tail -2 $O/step.txt
