"""Micro-benchmark of single conv shapes (1x1 mostly): ours with / without residual vs torch.matmul and plain streaming kernels."""
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
shapes = [(16, 128, 128, 64, 256), (16, 64, 64, 128, 512), (16, 32, 32, 256, 1024), (16, 32, 32, 1024, 256), (16, 64, 64, 512, 128), (16, 16, 16, 512, 2048)]
for B, H, W, Cc, N in shapes:
    M = B * H * W
    segs = [ho.Seg(B, H, W, 0)]
    x = torch.randn(M, Cc, device='cuda').bfloat16()
    w = torch.randn(N, Cc, 1, 1, device='cuda') * 0.05
    wp = ho.pack_weight_fwd(w)
    res = torch.randn(M, N, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda')
    out = torch.empty_like(res)
    a = t(lambda: ho.conv2d_rows(x, segs, wp, N, 1, 1, 1, 0, 1, relu=True, pre_shift=bias, out=out))
    b = t(lambda: ho.conv2d_rows(x, segs, wp, N, 1, 1, 1, 0, 1, relu=True, pre_shift=bias, res=res, out=out))
    wt = w.reshape(N, Cc).t().contiguous().bfloat16()
    c = t(lambda: torch.matmul(x, wt, out=out))
    d = t(lambda: torch.add(res, res, out=out))
    mb = lambda r: (M * Cc + M * N * (2 if r else 1)) * 2 / 1e6
    print(f'M={M:7d} C={Cc:5d} N={N:5d}: ours {a:7.1f} us ({mb(0)/a*1e3:6.0f} GB/s)  +res {b:7.1f} us ({mb(1)/b*1e3:6.0f} GB/s)  matmul {c:7.1f} us ({mb(0)/c*1e3:6.0f} GB/s)  '
          f'add {d:7.1f} us ({3*M*N*2/1e6/d*1e3:6.0f} GB/s)')
