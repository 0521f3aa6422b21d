"""Micro-benchmark: head-tower conv (M=87296, N=256, K=2304) forward / dgrad / wgrad, N launches each."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
segs, r = [], 0
for h, w in sizes:
    segs.append(ho.Seg(B, h, w, r)); r += B * h * w
M = r
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.randn(M, 256, device='cuda').bfloat16()
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
dz = torch.randn(M, 256, device='cuda').bfloat16()
wp, wd = ho.pack_weight_fwd(w), ho.pack_weight_dgrad(w)
def timeit(fn, name, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f'{name:6s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF')
fl = 2.0 * M * 256 * 2304
if which in ('all', 'fwd'): timeit(lambda: ho.conv2d_rows(x, segs, wp, 256, 3, 3, 1, 1, 1, relu=True), 'fwd', fl)
if which == 'fwd1':
    w1 = torch.randn(256, 256, 1, 1, device='cuda') * 0.02
    wp1 = ho.pack_weight_fwd(w1)
    timeit(lambda: ho.conv2d_rows(x, segs, wp1, 256, 1, 1, 1, 0, 1, relu=True), 'fwd1', 2.0 * M * 256 * 256)
if which in ('all', 'dgrad'): timeit(lambda: ho.conv2d_dgrad_rows(dz, segs, segs, wd, 256, 3, 3, 1, 1, 1), 'dgrad', fl)
if which in ('all', 'wgrad'):
    def f():
        dw = ho.conv2d_wgrad_rows(x, segs, dz, segs, 3, 3, 1, 1, 1)
        ho.unpack_wgrad(dw, 256, 256)
    timeit(f, 'wgrad', fl)
