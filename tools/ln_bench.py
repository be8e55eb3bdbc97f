"""LayerNorm-over-channels forward / backward (dx) time per attention block of the Config D UNet (B = 256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tf_all = tb_all = 0.0
for name, (C, S) in [("sa1", (64, 16)), ("sa2", (128, 8)), ("sa3", (128, 4)), ("sa4", (64, 8)), ("sa5", (32, 16)), ("sa6", (32, 32))]:
    x = torch.randn(B, C, S, S, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    st = torch.empty(B * S * S, 2, device=dev); g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
    tf = bench.ev_time(lambda: L_.afd_layernorm_c_fwd(x.data_ptr(), y.data_ptr(), st.data_ptr(), B, C, S * S, 1e-5, g.data_ptr(), be.data_ptr(), s), reps=20, warm=3)
    tb = bench.ev_time(lambda: L_.afd_layernorm_c_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), B, C, S * S, g.data_ptr(), dx.data_ptr(), x.data_ptr(), None, None, None, 0, s), reps=20, warm=3)
    e = 4.0 * B * C * S * S
    tf_all += 2 * tf; tb_all += 2 * tb
    print(f"{name} C={C} {S}x{S}: fwd {tf*1e3:6.1f} us ({2*e/tf/1e6:6.0f} GB/s)  bwd dx {tb*1e3:6.1f} us ({4*e/tb/1e6:6.0f} GB/s)")
print(f"per step (2 LayerNorms per block): fwd {tf_all*1e3:.0f} us, bwd dx {tb_all*1e3:.0f} us")
