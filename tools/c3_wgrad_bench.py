import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
for (ci, co, S) in [(3, 32, 32), (3, 64, 32)]:
    x = torch.randn(B, ci, S, S, device=dev); dy = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    for mode in (85, 84):
        L.afd_debug_conv_path(mode)
        ws = torch.empty(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4 + 1, device=dev)
        f = lambda: L.afd_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
        f(); print(ci, co, S, "mode", mode, "form", L.afd_conv_wgrad_form(B, ci, co, S, S, 3), f"{bench.ev_time(f, reps=10)*1e3:.1f} us")
    L.afd_debug_conv_path(84)
