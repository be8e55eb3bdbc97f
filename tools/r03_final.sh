#!/bin/bash
# One GPU-box pass that produces what DESIGN.md section 6 cites for round 3 (run from the repo root on the MI355X box):
#   $O/{pytest.log,bench.json}, prof/ (step), fwdprof/ (sampling forward) = rocprofv3 --kernel-trace --stats,
#   pmc/{fetch,write,sq}/ (whole step), pmcf/<family>_{fetch,write,sq}/ (one conv family at a time), fwd/{fetch,write,sq1} (sampling).
# Counter passes never share a run with any trace domain other than --kernel-trace; the program after `--` is python3 itself.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_final}
STAGE=${2:-all}
mkdir -p $O
cd $R
if [ $STAGE = all ] || [ $STAGE = a ]; then
  python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest.log; tail -3 $O/pytest.log
  [ $rc -eq 0 ] || exit 1
  python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  cd /tmp && export TMPDIR=/tmp
  A="--steps 10 --warmup 3 --no-graph --no-sample --no-cpu-baseline --no-kernels"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python3 $R/bench.py $A > $O/prof.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwdprof -o fwd -- python3 $R/tools/fwd_profile.py > $O/fwdprof.log 2>&1 || exit 1
  echo "stage a done"
fi
if [ $STAGE = all ] || [ $STAGE = b ]; then
  cd /tmp && export TMPDIR=/tmp
  A="--steps 3 --warmup 1 --no-graph --no-sample --no-cpu-baseline --no-kernels"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -o f -- python3 $R/bench.py $A > $O/pmc_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -o w -- python3 $R/bench.py $A > $O/pmc_write.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/sq -o s -- python3 $R/bench.py $A > $O/pmc_sq.log 2>&1 || exit 1
  for f in fwd dgrad wgrad; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf/${f}_fetch -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_fetch.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcf/${f}_write -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_write.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcf/${f}_sq -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_sq.log 2>&1 || exit 1
  done
  for p in fetch:FETCH_SIZE write:WRITE_SIZE sq1:"SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    n=${p%%:*}; c=${p#*:}
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fwd/$n -o p -- python3 $R/tools/fwd_profile.py > $O/fwd_$n.log 2>&1 || exit 1
  done
  echo "stage b done"
fi
