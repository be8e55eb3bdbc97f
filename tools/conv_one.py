"""Run one conv shape repeatedly (for rocprofv3 --pmc): python tools/conv_one.py ci co S mode reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
ci, co, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
B = 256
dev = torch.device("cuda:0")
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
for _ in range(reps):
    if mode == "fwd":
        L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s)
    elif mode == "dgrad":
        L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s)
    else:
        L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
torch.cuda.synchronize()
