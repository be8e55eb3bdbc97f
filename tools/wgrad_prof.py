"""wgrad only, a few shapes (B=256), for rocprofv3 --kernel-trace --stats: splits main kernel vs slab reduce."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = [(3, 32, 32), (32, 32, 32), (64, 64, 32), (32, 32, 16), (128, 128, 16), (64, 64, 8), (256, 256, 8), (128, 128, 4), (256, 256, 4)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (ci, co, S) in shapes:
    x = torch.randn(B, ci, S, S, device=dev); y = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
    torch.cuda.synchronize()
    for _ in range(5):
        L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
    torch.cuda.synchronize()
    print("done", ci, co, S, flush=True)
