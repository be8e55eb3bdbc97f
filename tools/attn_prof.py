"""One launch of each attention kernel (d=8, L=1024, B=256) for counter collection under rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); B, C, S = 256, 32, 32
L_, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
Lq = S * S
qkv = torch.randn(B, 3 * C, S, S, device=dev); o = torch.empty(B, C, S, S, device=dev)
lse = torch.empty(B, 4, Lq, device=dev); dq = torch.empty_like(qkv); dl = torch.empty_like(lse)
for r in (0, 2):
    L_.afd_debug_attn_rows(r)
    for _ in range(2):
        L_.afd_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, 4, C // 4, Lq, s)
        L_.afd_attn_bwd(qkv.data_ptr(), o.data_ptr(), o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), B, 4, C // 4, Lq, s)
torch.cuda.synchronize()
