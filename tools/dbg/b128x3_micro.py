"""The fused 128-plane identity bottleneck of the reference-precision mode (aod_bottleneck128x3_fwd) against its three launches: equal bits,
and microseconds per block at the bench shape (16 x 64 x 64), inference and training-forward (kept intermediates) forms.
    gpurun -- 'python tools/dbg/b128x3_micro.py'"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import functional as AF      # noqa: E402
from aod_meh_hua_amd import hipops as ho          # noqa: E402
from aod_meh_hua_amd.models.backbones.resnet import Bottleneck      # noqa: E402


def main():
    AF.set_precision('bf16x3')
    B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 64, 64)
    g = torch.Generator(device='cuda').manual_seed(3)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    blk = Bottleneck(512, 128).cuda().eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(rnd(*m.weight.shape) / (m.weight[0].numel()) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, device='cuda', generator=g) + 0.5); m.bias.copy_(rnd(*m.bias.shape) * 0.1)
                m.running_mean.copy_(rnd(*m.bias.shape) * 0.1); m.running_var.copy_(torch.rand(m.bias.shape, device='cuda', generator=g) + 0.5)
    for q in blk.parameters():
        q.requires_grad_(False)
    x = AF.as_nchw(ho.x3_split(rnd(B * H * W, 512).relu()), B, H, W)

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res = {}
    for fused in ('1', '0'):
        os.environ['AOD_FUSE_BOTTLENECK128_X3'] = fused
        with torch.no_grad():
            y = blk(x)
            res[fused] = (y.clone(), timed(lambda: blk(x)))
    print(f'{B}x{H}x{W} inference: fused {res["1"][1]:.1f} us, three launches {res["0"][1]:.1f} us, equal bits: {torch.equal(res["1"][0], res["0"][0])}')
    bn = lambda n: (n.weight, n.bias, n.running_mean, n.running_var)
    p1 = AF.PREP.get(blk.conv1.weight, bn(blk.norm1), 1024, blk.norm1.eps)
    p2 = AF.PREP.get(blk.conv2.weight, bn(blk.norm2), 256, blk.norm2.eps)
    p3 = AF.PREP.get(blk.conv3.weight, bn(blk.norm3), 256, blk.norm3.eps)
    rows = AF.as_rows(x)
    f = lambda keep: ho.bottleneck128_fwd(rows, B, H, W, p1.wf, p1.scale, p1.shift, p2.wf, p2.scale, p2.shift, p3.wf, p3.scale, p3.shift, keep=keep)
    print(f'kernel alone: {timed(lambda: f(False)):.1f} us; with the two intermediates stored: {timed(lambda: f(True)):.1f} us')


if __name__ == '__main__':
    main()
