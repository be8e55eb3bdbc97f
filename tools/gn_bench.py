"""GroupNorm(1, C) backward (plane sums + apply) time per normalisation site shape of the Config D UNet (B = 256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tot = 0.0
for (C, S) in sorted(set(bench.ACT_SITES), key=lambda t: (-t[1], t[0])):
    cnt = bench.ACT_SITES.count((C, S))
    x = torch.randn(B, C, S, S, device=dev); dy = torch.randn_like(x); dx = torch.empty_like(x)
    st = torch.zeros(B, 2, device=dev); st[:, 1] = 1
    g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev); part = torch.empty(B, 2, C, device=dev)
    L_.afd_groupnorm1_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), B, C, S * S, g.data_ptr(), be.data_ptr(), None, 0, dx.data_ptr(), None, part.data_ptr(), None, 0, None, None, 0, s)
    t = bench.ev_time(lambda: L_.afd_groupnorm1_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), B, C, S * S, g.data_ptr(), be.data_ptr(), None, 0, dx.data_ptr(), None, part.data_ptr(), None, 1, None, None, 0, s), reps=20, warm=3)
    e = 4.0 * B * C * S * S
    tot += t * cnt
    print(f"C={C:4d} {S:2d}x{S:<2d} x{cnt}: apply {t*1e3:6.1f} us ({3*e/t/1e6:6.0f} GB/s)")
print(f"gn_bwd_apply over the 22 filtered sites: {tot*1e3:.0f} us per step")
