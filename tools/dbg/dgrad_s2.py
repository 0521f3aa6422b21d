"""Micro-benchmark: dgrad of stride-2 convs (class-major order + dead-tap skipping; AOD_DGRAD_CLASSES=0 turns it off)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, H, W, Cin, N, R) in [(16, 64, 64, 512, 1024, 1), (16, 64, 64, 256, 256, 3), (16, 128, 128, 128, 128, 3)]:
    OH, OW = H // 2, W // 2
    xs, zs = [ho.Seg(B, H, W)], [ho.Seg(B, OH, OW)]
    dz = torch.randn(B * OH * OW, N, device='cuda').bfloat16()
    w = torch.randn(N, Cin, R, R, device='cuda') * 0.05
    wd = ho.pack_weight_dgrad(w)
    out = torch.empty(B * H * W, Cin, device='cuda', dtype=torch.bfloat16)
    mask = torch.randn(B * H * W, Cin, device='cuda').bfloat16()
    res = torch.randn(B * H * W, Cin, device='cuda').bfloat16()
    cs = torch.zeros(Cin, device='cuda')
    a = t(lambda: ho.conv2d_dgrad_rows(dz, zs, xs, wd, Cin, R, R, 2, R // 2, 1, out=out))
    b = t(lambda: ho.conv2d_dgrad_rows(dz, zs, xs, wd, Cin, R, R, 2, R // 2, 1, out=out, mask=mask, res=res, colsum=cs))
    print(f'dX {B}x{H}x{W}x{Cin} from dZ x{N}, {R}x{R} s2: plain {a:7.1f} us   +res+mask+colsum {b:7.1f} us')
