#!/bin/bash
# Round-3 first GPU pass: parity suite, the bench line, the sampling-forward kernel trace, aten copy sites.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run1}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python tools/aten_copy_sites.py > $O/copy_sites.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwdprof -o fwd -- python3 $R/tools/fwd_profile.py > $O/fwdprof.log 2>&1 || exit 1
A="--steps 10 --warmup 3 --no-graph --no-sample --no-cpu-baseline --no-kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python3 $R/bench.py $A > $O/prof.log 2>&1 || exit 1
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline'],d['sample']['value'],d['sample'].get('roofline'))"
