"""Winograd vs direct 3x3 convolution per Config-D layer shape (B=256): forward and dgrad, microseconds and
algorithmic TFLOP/s (2*9*Cin*Cout per output pixel), forced 64- and 32-channel Winograd workgroups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afdm
import bench

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
print(f"{'shape':>20} {'n':>2} {'GFLOP':>6} | {'dir fwd':>8} {'64x64':>8} {'32x64':>8} {'64x32':>8} {'splitK':>8} {'auto':>8} | {'dir dgr':>8} {'64x64':>8} {'32x64':>8} {'64x32':>8} {'splitK':>8} {'auto':>8}   (us)")
tot = {"dir": 0.0, "auto": 0.0}
for (ci, co, S) in shapes:
    if ci < 8:
        continue
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x)
    u = torch.empty(16 * ci * co, device=dev)
    fl = 2.0 * B * S * S * ci * co * 9
    row = []
    for dgrad in (0, 1):
        def direct():
            if dgrad: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s)
            else: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s)
        def wino():
            if dgrad: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
            else: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
        td = bench.ev_time(direct, reps=10)
        ts = []
        for mode in (66, 67, 68, 70, 64):
            L.afd_debug_conv_path(mode)
            if L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, dgrad):
                ts.append(bench.ev_time(wino, reps=10))
            else:
                ts.append(float("nan"))
        L.afd_debug_conv_path(64)
        auto = ts[4] if ts[4] == ts[4] else td
        tot["dir"] += td * cnt; tot["auto"] += auto * cnt
        row.append((td, ts))
    f = lambda t: f"{t * 1e3:8.1f}"
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} {cnt:2d} {fl / 1e9:6.2f} | {f(row[0][0])} {' '.join(f(t) for t in row[0][1])} | "
          f"{f(row[1][0])} {' '.join(f(t) for t in row[1][1])}")
print("fwd+dgrad per step over these layers: direct %.3f ms, auto rule %.3f ms" % (tot["dir"], tot["auto"]))
