"""Runs ONE kernel family of the Config-D train step in isolation -- every 3x3 layer's forward, dgrad or wgrad, through
the same dispatch as ops.py -- `reps` times, so that a rocprofv3 --pmc pass over this process yields the family's HBM
traffic per step (tools/pmc_summary.py sums the counters over the process).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o f -- python3 tools/conv_family.py wgrad 3
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench

fam = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
bufs = {}
for (ci, co, S) in bench.CONV3:
    if (ci, co, S) in bufs:
        continue
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
    nf, nd = L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 0), L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 1)
    u = torch.empty(max(nf, nd, 4) // 4, device=dev)
    bufs[(ci, co, S)] = (x, w, y, dx, dw, ws, u, nf, nd)
torch.cuda.synchronize()
for _ in range(reps):
    for (ci, co, S) in bench.CONV3:
        x, w, y, dx, dw, ws, u, nf, nd = bufs[(ci, co, S)]
        if fam == "fwd":
            if nf: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
            else: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s)
        elif fam == "dgrad":
            if ci <= 3: continue
            if nd: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
            else: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s)
        else:
            L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
torch.cuda.synchronize()
print(f"{fam}: {reps} x {len(bench.CONV3)} layers")
