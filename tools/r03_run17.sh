#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run17}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwdprof -o fwd -- python3 $R/tools/fwd_profile.py > $O/fwdprof.log 2>&1 || exit 1
