"""Repeatability screen of the fused 128-plane x3 block: fused vs itself, three launches vs themselves, fused vs three launches, many runs per
shape; prints where mismatches sit (pixel, channel, tile coordinates).
    gpurun -- 'python tools/dbg/b128x3_race.py'"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import functional as AF      # noqa: E402
from aod_meh_hua_amd import hipops as ho          # noqa: E402
from aod_meh_hua_amd.models.backbones.resnet import Bottleneck      # noqa: E402

AF.set_precision('bf16x3')
g = torch.Generator(device='cuda').manual_seed(23)
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
N = int(os.environ.get('REPS', 30))
for (B, H, W) in ((1, 13, 37), (2, 16, 32), (3, 7, 129), (16, 64, 64), (2, 21, 37)):
    x = AF.as_nchw(ho.x3_split(rnd(B * H * W, 512)), B, H, W)

    def run(fused):
        os.environ['AOD_FUSE_BOTTLENECK128_X3'] = '1' if fused else '0'
        with torch.no_grad():
            return AF.as_rows(blk(x)).clone()
    f0, u0 = run(True), run(False)
    torch.cuda.synchronize()
    bad = dict(ff=0, uu=0, fu=0)
    where = []
    for it in range(N):
        f, u = run(True), run(False)
        torch.cuda.synchronize()
        bad['ff'] += int(not torch.equal(f, f0)); bad['uu'] += int(not torch.equal(u, u0)); bad['fu'] += int(not torch.equal(f, u))
        if not torch.equal(f, u) and len(where) < 3:
            d = (f.float() - u.float()).abs()
            idx = d.nonzero()
            rows = idx[:, 0].unique().tolist()[:8]
            where.append((int(idx.shape[0]), [(r // W % H, r % W) for r in rows], sorted(set((idx[:, 1] // 64).tolist()))[:12], float(d.max())))
    print(f'{B}x{H}x{W}: runs {N}: fused != fused {bad["ff"]}, unfused != unfused {bad["uu"]}, fused != unfused {bad["fu"]}; first f0==u0 {torch.equal(f0, u0)}', where)
