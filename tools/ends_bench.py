"""The network's ends at B = 256: first 3x3 layer forward (3 -> 32 @32x32, tiled kernel either way), output 1x1 layer forward / dgrad (32 -> 3):
streaming vector kernels (csrc/ends.hip) against the tiled kernels, us per launch and GB/s of the algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
x3 = torch.randn(B, 3, 32, 32, device=dev); w3 = torch.randn(32, 3, 3, 3, device=dev) * 0.1; y32 = torch.empty(B, 32, 32, 32, device=dev)
x32 = torch.randn(B, 32, 32, 32, device=dev); w1 = torch.randn(3, 32, 1, 1, device=dev) * 0.1; b1 = torch.randn(3, device=dev); y3 = torch.empty(B, 3, 32, 32, device=dev)
dx32 = torch.empty_like(x32)
by = 4.0 * B * 1024 * 35
for tag, mode in (("tiled", 93), ("streaming", 92)):
    L.afd_debug_conv_path(mode)
    t1 = bench.ev_time(lambda: L.afd_conv_fwd(P(x3), P(w3), None, None, P(y32), B, 3, 32, 32, 32, 3, 0, s), reps=10)
    t2 = bench.ev_time(lambda: L.afd_conv_fwd(P(x32), P(w1), P(b1), None, P(y3), B, 32, 3, 32, 32, 1, 0, s), reps=10)
    t3 = bench.ev_time(lambda: L.afd_conv_dgrad(P(y3), P(w1), P(dx32), B, 32, 3, 32, 32, 1, s), reps=10)
    print(f"{tag:10s}: first layer fwd {t1*1e3:6.1f} us ({by/t1/1e6:5.0f} GB/s) | output layer fwd {t2*1e3:6.1f} us ({by/t2/1e6:5.0f} GB/s) | its dgrad {t3*1e3:6.1f} us ({by/t3/1e6:5.0f} GB/s)")
L.afd_debug_conv_path(92)
