#!/bin/bash
# Stall-oriented PMC passes over the 3x3 forward family (tools/conv_family.py fwd): what the waves of conv_h2 wait for.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_stall}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT"
P3="TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P4="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL"
for p in p1:"$P1" p2:"$P2" p3:"$P3" p4:"$P4"; do
  n=${p%%:*}; c=${p#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o p -- python3 $R/tools/conv_family.py fwd 3 > $O/$n.log 2>&1 || { tail -5 $O/$n.log; echo "pass $n failed"; }
done
cd $R
python3 tools/pmc_kernel_table.py 3 $O/p1 $O/p2 $O/p3 $O/p4 > $O/table.json
