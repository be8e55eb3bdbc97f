#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3_run12}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention or silu or linear_wgrad or gelu_maxpool" > $O/pytest_attn.log 2>&1; rc=$?; tail -6 $O/pytest_attn.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" $O/pytest_attn.log | tail -30; exit 1; }
python - <<'PY' > $O/attn_ab.txt 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
for mode in (20, 21, 20, 21):
    L_.afd_debug_attn_rows(mode)
    out = []
    for name, (C, S) in [("sa5", (32, 16)), ("sa6", (32, 32))]:
        Lq = S * S
        qkv = torch.randn(B, 3 * C, S, S, device=dev); o = torch.empty(B, C, S, S, device=dev)
        lse = torch.empty(B, 4, Lq, device=dev); dq = torch.empty_like(qkv); dl = torch.empty_like(lse)
        tf = bench.ev_time(lambda: L_.afd_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, 4, C // 4, Lq, s), reps=10, warm=2)
        tb = bench.ev_time(lambda: L_.afd_attn_bwd(qkv.data_ptr(), o.data_ptr(), o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), B, 4, C // 4, Lq, s), reps=10, warm=2)
        out.append(f"{name}: fwd {tf*1e3:7.1f} bwd {tb*1e3:7.1f} us")
    print("mode", mode, " | ".join(out), flush=True)
PY
cat $O/attn_ab.txt | tail -5
