#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run10}
mkdir -p $O
cd $R
for v in base il base il; do
  if [ $v = il ]; then export AFD_LIBPATH=$R/tools/micro/bin/libafd_il.so; else unset AFD_LIBPATH; fi
  timeout -k 10 200 python tools/h2_bench.py 2>&1 | grep "per step (ms), direct" | sed "s/^/$v: /"
done
