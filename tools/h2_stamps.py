"""Phase timeline of the LDS-fed 3x3 kernel from in-kernel s_memtime stamps (library built by `tools/abl_conv.sh 0_stamp`,
AFD_LIBPATH=tools/micro/bin/libafd_h2abl_0_stamp.so): per workgroup 16 stamps through the `res` argument.
  python tools/h2_stamps.py [Cin Cout S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
import numpy as np
dev = torch.device("cuda:0"); B = 256
ci, co, S = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 64, 32)
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
y = torch.empty(B, co, S, S, device=dev); u = torch.empty(16 * ci * co, device=dev)
kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
stamps = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, stamps.data_ptr(), y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, kinds, s)
torch.cuda.synchronize(); stamps.zero_()
L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, stamps.data_ptr(), y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, kinds, s)
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 16)
st = st[st[:, 0] != 0]
t0 = st[:, 0].min()
names = ["start", "loads issued", "plan done", "x0 landed", "X0 committed", "row0", "row1", "row2", "x1 landed", "X1 committed", "row0", "row1", "row2", "loop done", "stores issued"]
print(f"{ci}->{co} @{S}: {len(st)} workgroups; ")
d = np.diff(st[:, :15].astype(np.int64), axis=1)
for i in range(14):
    col = d[:, i]
    if ci < 64 and 7 <= i <= 12:
        continue
    print(f"  {names[i]:>14s} -> {names[i+1]:<14s} mean {col.mean():8.0f}  p10 {np.percentile(col, 10):8.0f}  p90 {np.percentile(col, 90):8.0f}")
life = st[:, 14] - st[:, 0]
print(f"  workgroup lifetime mean {life.mean():.0f}  p10 {np.percentile(life,10):.0f} p90 {np.percentile(life,90):.0f}")
