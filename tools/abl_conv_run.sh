#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in base "$@" base; do
  if [ $v = base ]; then unset AFD_LIBPATH; else export AFD_LIBPATH=$R/tools/micro/bin/libafd_h2abl_$v.so; fi
  echo -n "$v: "; timeout -k 10 120 python tools/abl_conv_bench.py 2>&1 | grep total || exit 1
done
