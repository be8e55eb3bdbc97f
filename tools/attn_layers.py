"""Isolated attention-core time per attention block of the Config D UNet (B = 256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tot_f = tot_b = 0.0
for name, (C, S) in [("sa1", (64, 16)), ("sa2", (128, 8)), ("sa3", (128, 4)), ("sa4", (64, 8)), ("sa5", (32, 16)), ("sa6", (32, 32))]:
    Lq = S * S
    qkv = torch.randn(B, 3 * C, S, S, device=dev); o = torch.empty(B, C, S, S, device=dev)
    lse = torch.empty(B, 4, Lq, device=dev); dq = torch.empty_like(qkv); dl = torch.empty_like(lse)
    tf = bench.ev_time(lambda: L_.afd_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, 4, C // 4, Lq, s), reps=10, warm=2)
    tb = bench.ev_time(lambda: L_.afd_attn_bwd(qkv.data_ptr(), o.data_ptr(), o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), B, 4, C // 4, Lq, s), reps=10, warm=2)
    fl = 4.0 * Lq * Lq * C * B
    tot_f += tf; tot_b += tb
    print(f"{name} C={C} L={Lq} d={C//4}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:6.1f} TFLOP/s)  bwd {tb*1e3:7.1f} us ({2.5*fl/tb/1e9:6.1f} TFLOP/s)")
print(f"total fwd {tot_f*1e3:.0f} us, bwd {tot_b*1e3:.0f} us")
