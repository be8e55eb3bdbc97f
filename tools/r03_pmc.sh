#!/bin/bash
# PMC passes of round 3: the sampling forward (whole denoise steps) and the three conv families in isolation.
# Counter passes never share a run with any trace domain other than --kernel-trace; the program after `--` is python3 itself.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
for p in fetch:FETCH_SIZE write:WRITE_SIZE sq1:"$SQ1" sq2:"$SQ2"; do
  n=${p%%:*}; c=${p#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fwd/$n -o p -- python3 $R/tools/fwd_profile.py > $O/fwd_$n.log 2>&1 || { tail -5 $O/fwd_$n.log; exit 1; }
  echo "fwd $n done"
done
for f in wgrad fwd; do
  for p in fetch:FETCH_SIZE write:WRITE_SIZE sq1:"$SQ1" sq2:"$SQ2"; do
    n=${p%%:*}; c=${p#*:}
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/conv_$f/$n -o p -- python3 $R/tools/conv_family.py $f 3 > $O/conv_${f}_$n.log 2>&1 || { tail -5 $O/conv_${f}_$n.log; exit 1; }
  done
  echo "conv $f done"
done
cd $R
python3 tools/pmc_kernel_table.py 3 $O/conv_wgrad/sq1 $O/conv_wgrad/sq2 $O/conv_wgrad/fetch $O/conv_wgrad/write > $O/conv_wgrad_table.json
python3 tools/pmc_kernel_table.py 3 $O/conv_fwd/sq1 $O/conv_fwd/sq2 $O/conv_fwd/fetch $O/conv_fwd/write > $O/conv_fwd_table.json
python3 tools/pmc_kernel_table.py 30 $O/fwd/sq1 $O/fwd/sq2 $O/fwd/fetch $O/fwd/write > $O/fwd_table.json
ls -la $O
