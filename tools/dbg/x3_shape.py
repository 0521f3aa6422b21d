"""one reference-precision conv shape, timed:  python tools/dbg/x3_shape.py B H W Cin Cout R [stride]   (env knobs: AOD_TILE_WANT, AOD_TILE_256, ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aod_meh_hua_amd import functional as AF, hipops as ho
AF.set_precision('bf16x3')
B, H, W, Cc, N, R = map(int, sys.argv[1:7])
st = int(sys.argv[7]) if len(sys.argv) > 7 else 1
M = B * H * W
segs = [ho.Seg(B, H, W, 0)]
x = ho.x3_split(torch.randn(M, Cc, device='cuda'))
w = torch.randn(N, Cc, R, R, device='cuda') * 0.05
wp = ho.x3_split(w.permute(0, 2, 3, 1).reshape(N * R * R, Cc).contiguous()).view(N, R, R, -1)
bias = torch.randn(N, device='cuda')
f = lambda: ho.conv2d_rows(x, segs, wp, N, R, R, st, R // 2, 1, relu=True, pre_shift=bias)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 30 * 1e3
oh, ow = ho.out_hw(H, W, R, R, st, R // 2, 1)
print(f'{sys.argv[1:]} want={os.environ.get("AOD_TILE_WANT")}: {us:7.1f} us  {2.0 * B * oh * ow * N * Cc * R * R / us / 1e6:6.1f} TF alg', flush=True)
