#!/bin/bash
# Stall-oriented PMC passes over the F4 family (tools/f4_family.py): what the waves of filt_act_{fwd,bwd}_n3 wait for.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_stall_f4}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
for p in p1:"$P1" p2:"$P2"; do
  n=${p%%:*}; c=${p#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o p -- python3 $R/tools/f4_family.py 3 > $O/$n.log 2>&1 || { tail -5 $O/$n.log; echo "pass $n failed"; }
done
cd $R
python3 tools/pmc_kernel_table.py 3 $O/p1 $O/p2 > $O/table.json
