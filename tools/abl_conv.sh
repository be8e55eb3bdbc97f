#!/bin/bash
# Ablation libraries of the f16x2 3x3 kernel: builds tools/micro/bin/libafd_h2abl_<mask>[_aq<n>].so (the product library with h2.hip
# compiled under -DAFD_H2_ABL=<mask> / -DAFD_H2_AQ=<n>); run on the box with AFD_LIBPATH (tools/abl_conv_run.sh).  Timings only.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/aliasfree-diffusion-models-pytorch_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $R/tools/micro/bin
for v in "$@"; do
  m=${v%%_*}; aq=2; pr=0; [[ $v == *_aq* ]] && aq=${v##*_aq}; [[ $v == *_pr* ]] && pr=${v##*_pr}; sl=0; [[ $v == *_sl* ]] && sl=${v##*_sl}; st=0; [[ $v == *_stamp* ]] && st=1; pad=0; [[ $v == *_pad* ]] && pad=40000
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DAFD_H2_ABL=$m -DAFD_H2_AQ=$aq -DAFD_H2_PRIO=$pr -DAFD_H2_SLEEP=$sl -DAFD_H2_STAMP=$st -DAFD_H2_LDSPAD=$pad -c $C/h2.hip -o /tmp/h2_abl_$v.o
  objs=$(ls $C/build/*.o | grep -v "/h2.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/h2_abl_$v.o -o $R/tools/micro/bin/libafd_h2abl_$v.so
  echo built $v
done
