"""Stem timing: 7x7 / stride-2 conv over the 8-channel NHWC image vs the 4x4 / stride-1 conv over the space-to-depth image (+ layout kernels, + max-pool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import functional as AF, hipops as ho
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
dev = torch.device('cuda')
model, cfg = B.build_model(dev, B.CONFIGS['voc512'])
bb = model.backbone
img = torch.randn(16, 3, 512, 512, device=dev)
with torch.no_grad():
    x8 = AF.image_to_nhwc(img, 8)
    print('nchw->nhwc8   %.1f us' % t(lambda: AF.image_to_nhwc(img, 8)))
    print('conv 7x7 s2   %.1f us' % t(lambda: bb.conv1(x8, bn=bb.norm1, relu=True)))
    print('nchw->s2d     %.1f us' % t(lambda: ho.nchw_to_s2d_rows(img)))
    print('s2d stem all  %.1f us' % t(lambda: AF.stem_conv_s2d(img, bb.conv1, bb.norm1)))
    y = AF.stem_conv_s2d(img, bb.conv1, bb.norm1)
    print('maxpool       %.1f us' % t(lambda: AF.max_pool_3x3_s2(y)))
    print('fused s2d + conv + bn + relu + pool  %.1f us' % t(lambda: AF.stem_pool_s2d(img, bb.conv1, bb.norm1)))
    for shp in ((16, 512, 512), (2, 128, 160), (3, 62, 34), (1, 30, 66)):
        im = torch.randn(shp[0], 3, shp[1], shp[2], device=dev)
        a = AF.stem_pool_s2d(im, bb.conv1, bb.norm1)
        b = AF.max_pool_3x3_s2(AF.stem_conv_s2d(im, bb.conv1, bb.norm1))
        torch.cuda.synchronize()
        print(shp, 'fused == separate:', a.shape == b.shape and bool(torch.equal(a, b)), 'max|diff|', float((a.float() - b.float()).abs().max()), 'mean', float(b.float().abs().mean()))
