"""A/B of the two fwd/dgrad tilings per small layer: path 1 = 64x128 tile, 2 = 32x64 split-K tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = [t for t in sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1])) if t[2] <= 16 and t[0] >= 32]
for (ci, co, S) in shapes:
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x)
    fl = 2.0 * B * S * S * ci * co * 9
    out = []
    for mode in (1, 2):
        L.afd_debug_conv_path(mode)
        tf = bench.ev_time(lambda: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s), reps=10)
        td = bench.ev_time(lambda: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s), reps=10)
        out.append((tf, td))
    L.afd_debug_conv_path(0)
    P = B * S * S
    print(f"{ci:4d}->{co:4d} @{S:2d}  wgs_big={((P+127)//128)*((co+63)//64):5d} | big fwd {out[0][0]*1e3:7.1f}us {fl/out[0][0]/1e9:5.1f}TF dgrad {out[0][1]*1e3:7.1f}us | sk fwd {out[1][0]*1e3:7.1f}us {fl/out[1][0]/1e9:5.1f}TF dgrad {out[1][1]*1e3:7.1f}us")
