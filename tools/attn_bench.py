"""Attention timing per block (B=256) for rows-per-lane variants (d=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
for (C, S) in [(32, 32), (32, 16), (64, 16), (128, 8)]:
    Lq = S * S
    qkv = torch.randn(B, 3 * C, S, S, device=dev); o = torch.empty(B, C, S, S, device=dev)
    lse = torch.empty(B, 4, Lq, device=dev); dq = torch.empty_like(qkv); dl = torch.empty_like(lse)
    for r in ((0, 1, 2, 4) if C == 32 else (0,)):
        L_.afd_debug_attn_rows(r)
        tf = bench.ev_time(lambda: L_.afd_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, 4, C // 4, Lq, s), reps=5, warm=1)
        tb = bench.ev_time(lambda: L_.afd_attn_bwd(qkv.data_ptr(), o.data_ptr(), o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), B, 4, C // 4, Lq, s), reps=5, warm=1)
        print(f"C={C} L={Lq} d={C//4} R={r}: fwd {tf*1e3:8.1f} us  bwd {tb*1e3:8.1f} us")
    L_.afd_debug_attn_rows(0)
    if C == 32:
        for f in (10, 11):
            L_.afd_debug_attn_rows(f)
            tb = bench.ev_time(lambda: L_.afd_attn_bwd(qkv.data_ptr(), o.data_ptr(), o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), B, 4, C // 4, Lq, s), reps=5, warm=1)
            print(f"C={C} L={Lq} d={C//4} mfma8 bwd at every L={'on' if f == 11 else 'off'}: bwd {tb*1e3:8.1f} us")
        L_.afd_debug_attn_rows(10)
