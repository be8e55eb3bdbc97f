#!/bin/bash
# usage: abl_conv_run.sh "<modes>" variant...   (variants = tools/micro/bin/libafd_h2abl_<variant>.so, `base` = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
M=$1; shift
for v in base "$@" base; do
  if [ $v = base ]; then unset AFD_LIBPATH; else export AFD_LIBPATH=$R/tools/micro/bin/libafd_h2abl_$v.so; fi
  timeout -k 10 120 python tools/abl_conv_bench.py $M 2>&1 | grep total | sed "s/^/$v: /" || exit 1
done
