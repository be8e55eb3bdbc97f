"""Ablation of conv_wino: time with parts of the kernel switched off (results are wrong in those runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
L.afd_debug_conv_path(66)
for (ci, co, S) in [(64, 64, 32), (128, 128, 16), (256, 256, 8), (64, 32, 32)]:
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.empty(B, co, S, S, device=dev); u = torch.empty(16 * ci * co, device=dev)
    f = lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, s)
    out = []
    for name, d in [("full", 0), ("no weights kernel", 32), ("no mfma", 1 + 32), ("no transform", 2 + 32), ("no mfma/transform", 3 + 32),
                    ("no commit", 4 + 32), ("no fetch", 8 + 32), ("no commit/fetch", 12 + 32), ("no epilogue", 16 + 32),
                    ("only mfma", 2 + 4 + 8 + 16 + 32), ("only staging", 1 + 2 + 16 + 32), ("nothing", 31 + 32)]:
        L.afd_debug_conv_path(1000 + d)
        out.append(f"{name} {bench.ev_time(f, reps=10) * 1e3:.1f}")
    L.afd_debug_conv_path(1000)
    print(f"{ci}->{co}@{S}: " + " | ".join(out))
