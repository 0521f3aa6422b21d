"""reference-precision stem (image -> pooled X rows): time of the fused launch and of the three launches it replaces"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aod_meh_hua_amd import functional as AF
from aod_meh_hua_amd.mmcv_lite import BatchNorm2d, Conv2d
AF.set_precision('bf16x3')
conv = Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
bn = BatchNorm2d(64).cuda().eval()
for q in list(conv.parameters()) + list(bn.parameters()): q.requires_grad_(False)
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 512, 512)))
img = torch.randn(B, 3, H, W, device='cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for fuse in ('1', '0'):
    os.environ['AOD_STEM_POOL_FUSE'] = fuse
    print('fused' if fuse == '1' else 'three launches', f'{t(lambda: AF.stem_pool_s2d(img, conv, bn)):8.1f} us', flush=True)
