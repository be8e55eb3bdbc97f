"""3x3 forward (f16x2 tile kernel, weights ready) per Config-D layer shape at B = 256, one line: for the ablation libraries of
tools/abl_conv.sh (AFD_LIBPATH).  Microseconds per launch, then the per-step total over the layers the tile kernel takes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
for mode in [int(a) for a in sys.argv[1:]] or [None]:
  if mode is not None:
      L.afd_debug_conv_path(mode)
  out, tot = [f"mode {mode}:"], 0.0
  for (ci, co, S) in shapes:
      if ci < 32 or S < 8:
          continue
      kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
      if not kinds & 1:
          continue
      x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
      y = torch.empty(B, co, S, S, device=dev); u = torch.empty(16 * ci * co, device=dev)
      L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, kinds, s)
      t = bench.ev_time(lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, kinds, s), reps=20)
      tot += t * bench.CONV3.count((ci, co, S))
      out.append(f"{ci}>{co}@{S}:{t*1e3:.1f}")
  print(" ".join(out), f"| total {tot:.3f} ms", flush=True)
