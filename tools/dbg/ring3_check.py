"""Run convs that take the small 4-wave tiles (64 x 128, 128 x 64, 64 x 64: forward with BN + residual + ReLU, stride-2 forward, dgrad with
mask + column sums, K from 1 to 72 K-steps) under the current AOD_RING3 setting and save the results (tests/test_gpu_kernels.py compares
the three-stage ring with the two-stage loop)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
out_path = sys.argv[1]
torch.manual_seed(5)
res = {}
# (B, H, W, Cin, Cout, R, stride): tile counts chosen so that the dispatcher lands on the small tiles
cases = [(4, 32, 32, 256, 256, 3, 1), (4, 32, 32, 1024, 256, 1, 1), (2, 33, 29, 64, 192, 3, 1), (4, 32, 32, 256, 512, 3, 2), (3, 16, 16, 512, 2048, 1, 1),
         (4, 16, 16, 512, 512, 3, 1), (2, 19, 23, 128, 40, 3, 1), (5, 8, 8, 2048, 256, 1, 1), (1, 64, 64, 64, 64, 1, 1)]
for ci, (B, H, W, Ci, Co, R, st) in enumerate(cases):
    pad = R // 2
    segs = [ho.Seg(B, H, W, 0)]
    M = B * H * W
    x = torch.randn(M, Ci, device='cuda').bfloat16()
    w = torch.randn(Co, Ci, R, R, device='cuda') / (Ci * R * R) ** 0.5
    scale, shift = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda')
    dseg = ho.out_segs(segs, R, R, st, pad, 1)
    Mo = sum(s.rows for s in dseg)
    r = torch.randn(Mo, Co, device='cuda').bfloat16() if st == 1 else None
    y, _ = ho.conv2d_rows(x, segs, ho.pack_weight_fwd(w), Co, R, R, st, pad, 1, pre_scale=scale, pre_shift=shift, res=r, relu=True)
    res[f'y{ci}'] = y.cpu()
    if Co % 8 == 0:           # (stride 2: the class-major dgrad, whose K loop hops over the taps that cannot reach a tile)
        dz = torch.randn(Mo, Co, device='cuda').bfloat16()
        cs = torch.zeros(Ci, device='cuda')
        dx = ho.conv2d_dgrad_rows(dz, dseg, segs, ho.pack_weight_dgrad(w), Ci, R, R, st, pad, 1, mask=x, colsum=cs)
        res[f'dx{ci}'] = dx.cpu()
torch.cuda.synchronize()
torch.save(res, out_path)
print('saved', out_path, len(res))
