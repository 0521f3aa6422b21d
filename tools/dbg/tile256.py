"""Micro-benchmark: 3x3 256->256 conv (forward, bias + ReLU) on the FPN-P3 shape and on the level-batched head-tower shape;
run once with AOD_TILE_256=1 and once without.  Checks the result against the other tile's output saved by the first run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
cases = {'p3 65536': [(16, 64, 64)], 'tower 87296': [(16, 64, 64), (16, 32, 32), (16, 16, 16), (16, 8, 8), (16, 4, 4)], 'x2 131072': [(32, 64, 64)], 'x4 262144': [(64, 64, 64)]}
for name, lv in cases.items():
    segs, r0 = [], 0
    for b, h, w in lv:
        segs.append(ho.Seg(b, h, w, r0)); r0 += b * h * w
    M = r0
    x = torch.randn(M, 256, device='cuda').bfloat16()
    w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
    wp = ho.pack_weight_fwd(w)
    bias = torch.randn(256, device='cuda')
    out = torch.empty(M, 256, device='cuda', dtype=torch.bfloat16)
    fn = lambda: ho.conv2d_rows(x, segs, wp, 256, 3, 3, 1, 1, 1, relu=True, pre_shift=bias, out=out)
    us = t(fn)
    tag = 't256' if os.environ.get('AOD_TILE_256') == '1' else 't128'
    f = f'/tmp/tile256_{name.split()[0]}.pt'
    msg = ''
    if os.path.exists(f):
        ref = torch.load(f)
        msg = f'max |diff| vs other tile {float((out.float().cpu() - ref.float()).abs().max()):.4f} (bf16 ulp scale {float(ref.float().abs().max()) / 128:.4f})'
    else:
        torch.save(out.cpu(), f)
    print(f'{tag} {name:14s} {us:8.1f} us  {2.0 * M * 256 * 2304 / us / 1e6:8.1f} TFLOP/s  {msg}')
