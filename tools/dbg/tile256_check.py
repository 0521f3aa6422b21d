"""Run a level-batched 3x3 256->256 conv forward (bias + ReLU) and its dgrad (ReLU mask + column sums) with whatever tile the library
picks under the current AOD_TILE_256 setting and save the results (tests/test_gpu_kernels.py compares the 256 x 256 tile with the default)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
out_path = sys.argv[1]
torch.manual_seed(3)
segs, r0 = [], 0
for b, h, w in [(6, 64, 64), (6, 32, 32), (6, 16, 16), (6, 8, 8), (6, 4, 4)]:
    segs.append(ho.Seg(b, h, w, r0)); r0 += b * h * w
M = r0                                     # 32 736 rows: 128 tiles of 256 rows, the last one ragged
x = torch.randn(M, 256, device='cuda').bfloat16()
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
bias = torch.randn(256, device='cuda')
y, _ = ho.conv2d_rows(x, segs, ho.pack_weight_fwd(w), 256, 3, 3, 1, 1, 1, relu=True, pre_shift=bias)
dz = torch.randn(M, 256, device='cuda').bfloat16()
cs = torch.zeros(256, device='cuda')
dx = ho.conv2d_dgrad_rows(dz, segs, segs, ho.pack_weight_dgrad(w), 256, 3, 3, 1, 1, 1, mask=x, colsum=cs)
torch.cuda.synchronize()
torch.save(dict(y=y.cpu(), dx=dx.cpu(), cs=cs.cpu()), out_path)
print('saved', out_path, float(y.float().abs().mean()), float(dx.float().abs().mean()))
