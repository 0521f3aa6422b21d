"""A/B of the 256 x 256 wgrad tile (AOD_WGRAD_256=0 disables it) on the level-batched head-tower shape and the FPN P3 shape: run once
per setting; the second run compares its dW with the first run's."""
import os, sys, torch
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
torch.manual_seed(0)
cases = {'tower': [(16, 64, 64), (16, 32, 32), (16, 16, 16), (16, 8, 8), (16, 4, 4)], 'p3': [(16, 64, 64)], 'c4_3x3': [(16, 32, 32)], 'ragged': [(3, 37, 29), (3, 19, 15)]}
tag = 'w128' if os.environ.get('AOD_WGRAD_256') == '0' else 'w256'
for name, lv in cases.items():
    segs, r0 = [], 0
    for b, h, w in lv:
        segs.append(ho.Seg(b, h, w, r0)); r0 += b * h * w
    M = r0
    x = torch.randn(M, 256, device='cuda').bfloat16()
    dz = torch.randn(M, 256, device='cuda').bfloat16()
    zsegs = ho.out_segs(segs, 3, 3, 1, 1, 1)
    g = torch.empty(256, 256, 3, 3, device='cuda')
    def fn():
        slabs = ho.conv2d_wgrad_rows(x, segs, dz, zsegs, 3, 3, 1, 1, 1)
        ho.unpack_wgrad(slabs, 256, 256, grad_oihw=g)
        return slabs
    nsl = fn().shape[0]
    torch.cuda.synchronize()
    both = t(fn)
    only = t(lambda: ho.conv2d_wgrad_rows(x, segs, dz, zsegs, 3, 3, 1, 1, 1))
    f = f'/tmp/wgrad256_{name}.pt'
    msg = ''
    if os.path.exists(f):
        ref = torch.load(f)
        msg = f'max |diff| vs other tile {float((g.cpu() - ref).abs().max()):.3e} (|dW| max {float(ref.abs().max()):.1f})'
    else:
        torch.save(g.cpu(), f)
    print(f'{tag} {name:8s} M={M:6d} slabs {nsl:3d}  wgrad {only:7.1f} us ({2.0 * M * 256 * 2304 / only / 1e6:6.1f} TF)  wgrad+unpack {both:7.1f} us  {msg}', flush=True)
