#!/bin/bash
# One GPU-box pass that produces everything DESIGN.md section 6 cites (run from the repo root on the MI355X box):
#   gpurun_out/final/{pytest.log,bench.json}, prof/ (rocprofv3 --kernel-trace --stats), pmc/{fetch,write,sq}/ (whole step)
#   and pmcf/<family>_{fetch,write,sq}/ (one conv family at a time).  Counter passes never share a run with any trace
#   domain other than --kernel-trace.  The program after `--` is always python3 itself.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r02
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -2 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
A="--steps 10 --warmup 3 --no-graph --no-sample --no-cpu-baseline --no-kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python3 $R/bench.py $A > $O/prof.log 2>&1 || exit 1
A="--steps 3 --warmup 1 --no-graph --no-sample --no-cpu-baseline --no-kernels"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -o f -- python3 $R/bench.py $A > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -o w -- python3 $R/bench.py $A > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/sq -o s -- python3 $R/bench.py $A > $O/pmc_sq.log 2>&1 || exit 1
# the filtered-GELU family alone, with the wave-state counters (what bounds F4: profiles/r02_f4_pmc.json)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/f4/sq1 -o p -- python3 $R/tools/f4_family.py 3 > $O/f4_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/f4/sq2 -o p -- python3 $R/tools/f4_family.py 3 > $O/f4_sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f4/fetch -o p -- python3 $R/tools/f4_family.py 3 > $O/f4_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/f4/write -o p -- python3 $R/tools/f4_family.py 3 > $O/f4_write.log 2>&1 || exit 1
for f in fwd dgrad wgrad; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf/${f}_fetch -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcf/${f}_write -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_write.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcf/${f}_sq -o p -- python3 $R/tools/conv_family.py $f 3 > $O/pmcf_${f}_sq.log 2>&1 || exit 1
done
cat $O/bench.json
