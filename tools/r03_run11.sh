#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run11}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--steps 10 --warmup 3 --no-graph --no-sample --no-cpu-baseline --no-kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python3 $R/bench.py $A > $O/prof.log 2>&1 || exit 1
tail -2 $O/prof.log
