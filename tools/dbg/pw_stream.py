"""A/B of the persistent streaming pointwise kernel (csrc/pointwise.hip) against the general implicit-GEMM kernel on the 1x1 / stride-1
shapes of one bench step (R50, 16 x 512 x 512): bit-equality of the outputs and column sums, microseconds per launch of both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
from aod_meh_hua_amd._C import lib


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
# (B, H, W, C_in, C_out, kind): kind f = forward (BN scale/shift + ReLU), r = forward with residual, d = dgrad form (mask + column sums), dr = dgrad + residual
shapes = [(16, 128, 128, 64, 256, 'r'), (16, 128, 128, 256, 64, 'f'), (16, 128, 128, 64, 64, 'f'), (16, 128, 128, 256, 128, 'f'),
          (16, 64, 64, 128, 512, 'r'), (16, 64, 64, 512, 128, 'f'), (16, 64, 64, 512, 256, 'f'),
          (16, 32, 32, 256, 1024, 'r'), (16, 32, 32, 1024, 256, 'f'), (16, 32, 32, 1024, 512, 'f'),
          (16, 16, 16, 512, 2048, 'r'), (16, 16, 16, 2048, 512, 'f'), (16, 16, 16, 2048, 256, 'f'),
          (16, 128, 128, 256, 64, 'd'), (16, 128, 128, 64, 256, 'dr'), (16, 64, 64, 512, 128, 'd'), (16, 64, 64, 128, 512, 'dr'),
          (16, 32, 32, 1024, 256, 'd'), (16, 32, 32, 256, 1024, 'dr'), (16, 16, 16, 2048, 512, 'd'), (16, 16, 16, 512, 2048, 'dr'),
          (3, 37, 29, 64, 192, 'dr'), (2, 19, 23, 128, 64, 'r')]
only = sys.argv[1:]
tot = [0.0, 0.0]
for B, H, W, Ci, Co, kind in shapes:
    M = B * H * W
    segs = [ho.Seg(B, H, W, 0)]
    x = torch.randn(M, Ci, device='cuda').bfloat16()
    w = torch.randn(Co, Ci, 1, 1, device='cuda') * (1.0 / Ci ** 0.5)
    wp = ho.pack_weight_fwd(w)
    scale, shift = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda')
    res = torch.randn(M, Co, device='cuda').bfloat16() if 'r' in kind else None
    mask = torch.randn(M, Co, device='cuda').bfloat16() if 'd' in kind else None
    outs, sums, us = [], [], []
    for mode in (0, 1):
        lib.aod_set_pointwise_mode(mode)
        out = torch.full((M, Co), 7.0, device='cuda', dtype=torch.bfloat16)
        cs = torch.zeros(Co, device='cuda')
        if 'd' in kind:
            d = ho.make_desc(Ci, Co, 1, 1, 1, 0, 1, segs, segs, False, False, False)
            fn = lambda: ho.call('aod_conv2d', ho.C.byref(d), ho.ptr(x), ho.ptr(wp), ho.ptr(out), None, None, ho.ptr(res), ho.ptr(mask), None, None,
                                 ho.ptr(cs), ho.stream())
        else:
            fn = lambda: ho.conv2d_rows(x, segs, wp, Co, 1, 1, 1, 0, 1, pre_scale=scale, pre_shift=shift, res=res, relu=True, out=out)
        fn()
        torch.cuda.synchronize()
        outs.append(out.clone()); sums.append(cs.clone())
        us.append(t(fn))
    lib.aod_set_pointwise_mode(-1)
    same = bool((outs[0].view(torch.int16) == outs[1].view(torch.int16)).all())
    nbad = int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum())
    cerr = float((sums[0] - sums[1]).abs().max() / (sums[0].abs().max() + 1e-9))
    gb = 2.0 * M * (Ci + Co * (1 + ('r' in kind) + ('d' in kind))) / 1e3
    tot[0] += us[0]; tot[1] += us[1]
    print(f'{kind:2s} M={M:7d} K={Ci:5d} N={Co:5d}  general {us[0]:7.1f} us  stream {us[1]:7.1f} us  ({gb / us[1] / 1e3:5.2f} TB/s, {2.0 * M * Ci * Co / us[1] / 1e6:6.1f} TF)'
          f'  bit-equal {same} (differing {nbad})  colsum rel diff {cerr:.2e}', flush=True)
print(f'total general {tot[0]:.0f} us  stream {tot[1]:.0f} us')
