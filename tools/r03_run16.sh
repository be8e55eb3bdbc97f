#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run16}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest.log | tail -30; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --no-kernels > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['sample'])"
