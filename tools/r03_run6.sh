#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run6}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest.log | tail -30; exit 1; }
timeout -k 10 300 python tools/ab_modes.py "78" "79" > $O/ab_pw.txt 2>&1; tail -3 $O/ab_pw.txt
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['sample']['value'])
for r in d['kernels'][:24]: print(r['kernel'][:34].ljust(34), r['launches_per_step'], r['ms_per_step'], r.get('frac'), r.get('gbs'), r.get('tflops'))"
