#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run7}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" > $O/pytest_conv.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest_conv.log; tail -4 $O/pytest_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest_conv.log | tail -30; exit 1; }
timeout -k 10 300 python tools/h2_bench.py > $O/h2_bench.txt 2>&1; tail -14 $O/h2_bench.txt
timeout -k 10 300 python tools/ab_modes.py "74" "75" > $O/ab_sk.txt 2>&1; tail -3 $O/ab_sk.txt
