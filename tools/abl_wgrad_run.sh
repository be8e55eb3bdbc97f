#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in base vA vB vC; do
  if [ $v = base ]; then unset AFD_LIBPATH; else export AFD_LIBPATH=$R/tools/micro/bin/libafd_$v.so; fi
  timeout -k 10 200 python tools/h2_bench.py 2>&1 | grep "per step (ms), direct\|128-> 128 @16x16\|64->  64 @32x32" | sed "s/^/$v: /"
done
