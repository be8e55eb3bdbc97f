"""1x1 / Linear weight + bias gradient per attention-block shape (B = 256): the bf16x3 streaming kernel against the tiled
fp32-MFMA kernel, microseconds per call (kernel + slab reduction) and GB/s of the algorithmic bytes (x + dY once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tot = {"w": 0.0, "b": 0.0}
for (C, S) in [(32, 32), (64, 16), (128, 8)]:
    for co, cnt in ((3 * C, 1), (C, 3)):
        x = torch.randn(B, C, S, S, device=dev); dy = torch.randn(B, co, S, S, device=dev)
        dw = torch.empty(co, C, device=dev); db = torch.empty(co, device=dev)
        r = {}
        for tag, mode in (("w", 85), ("b", 84)):
            L.afd_debug_conv_path(mode)
            ws = torch.empty(L.afd_conv_wgrad_workspace_bytes(B, C, co, S, S, 1) // 4 + 1, device=dev)
            f = lambda: L.afd_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), B, C, co, S, S, 1, 0, ws.data_ptr(), s)
            f(); r[tag] = bench.ev_time(f, reps=10)
        L.afd_debug_conv_path(84)
        by = 4.0 * B * S * S * (C + co)
        print(f"{C:4d}->{co:4d} @{S:2d}x{S:<2d} x{2 * cnt}: tiled fp32 {r['w']*1e3:7.1f} us  bf16x3 {r['b']*1e3:7.1f} us ({by / r['b'] / 1e6:6.0f} GB/s)")
        for k in tot:
            tot[k] += 2 * cnt * r[k]
print("per step (ms): tiled %.3f  bf16x3 %.3f" % (tot["w"], tot["b"]))
