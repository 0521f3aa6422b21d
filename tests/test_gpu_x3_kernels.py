"""GPU: the reference-precision kernels (aod_conv_desc_t.x3, csrc/conv.hip "X3", csrc/x3_ops.hip) one by one against torch fp32 on the same
values.  An X-layout tensor carries 16 significant bits per value and a product drops the tail x tail term (2^-16), so the tolerances below
are 1e-4 of the result's scale -- two orders of magnitude under what the bf16 mode's tests allow (5e-2)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def x3_mode():
    from aod_meh_hua_amd import functional as AF
    AF.set_precision('bf16x3')
    yield
    AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))


def _x(t_nchw):
    """fp32 [B,C,H,W] -> X rows"""
    from aod_meh_hua_amd import hipops as ho
    B, C, H, W = t_nchw.shape
    return ho.x3_split(t_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())


def _f(rows, B, H, W, C):
    from aod_meh_hua_amd import hipops as ho
    return ho.x3_merge(rows, C).view(B, H, W, C).permute(0, 3, 1, 2)


def _err(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))


def test_split_merge_round_trip_keeps_16_bits():
    from aod_meh_hua_amd import hipops as ho
    g = torch.Generator(device='cuda').manual_seed(1)
    for C in (20, 64, 180):
        t = torch.randn(1000, C, device='cuda', generator=g) * 3
        x = ho.x3_split(t)
        assert x.shape == (1000, ho.xw(C)) and x.dtype == torch.bfloat16
        back = ho.x3_merge(x, C)
        assert float(((back - t).abs() / t.abs().clamp_min(1e-20)).max()) < 2 ** -15
        # pad channels are zero
        if C % 32:
            full = ho.x3_merge(x)
            assert float(full[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize('case', [
    dict(B=2, C=64, O=64, H=32, W=32, R=3, stride=1, pad=1, relu=True, bn=True, res=False),       # 64 x 64 / 64 x 128 tiles
    dict(B=2, C=256, O=64, H=24, W=40, R=1, stride=1, pad=0, relu=True, bn=True, res=False),
    dict(B=2, C=64, O=256, H=24, W=40, R=1, stride=1, pad=0, relu=True, bn=True, res=True),      # residual = head + tail
    dict(B=4, C=128, O=128, H=64, W=64, R=3, stride=2, pad=1, relu=True, bn=True, res=False),     # 128 x 128 tile, stride 2
    dict(B=2, C=512, O=256, H=8, W=8, R=3, stride=2, pad=1, relu=False, bn=False, res=False),     # small output, deep K (split-K)
    dict(B=2, C=2048, O=256, H=8, W=8, R=3, stride=2, pad=1, relu=False, bn=False, res=False),    # P6 shape: split-K
    dict(B=2, C=256, O=180, H=16, W=16, R=3, stride=1, pad=1, relu=False, bn=False, res=False, out_f32=True),     # prediction conv
    dict(B=2, C=256, O=9, H=16, W=16, R=3, stride=1, pad=1, relu=True, bn=False, res=False, out_f32=True),
    # tower shape, 51 842 pixels (not a multiple of the 32-pixel step): the 256 x 256 conv tile and the WIDE wgrad form (one accumulator per entry)
    dict(B=2, C=256, O=256, H=161, W=161, R=3, stride=1, pad=1, relu=True, bn=False, res=False),
    dict(B=1, C=512, O=256, H=224, W=224, R=1, stride=1, pad=0, relu=False, bn=True, res=False),     # wide wgrad of a 1x1 layer with BN
])
def test_x3_conv_forward_dgrad_wgrad_against_fp32(case):
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.mmcv_lite import BatchNorm2d
    c = dict(out_f32=False)
    c.update(case)
    B, C, O, H, W, R = c['B'], c['C'], c['O'], c['H'], c['W'], c['R']
    g = torch.Generator(device='cuda').manual_seed(5)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    x = rnd(B, C, H, W)
    w = (rnd(O, C, R, R) / (C * R * R) ** 0.5).requires_grad_()
    bn = None
    bias = None
    if c['bn']:
        bn = BatchNorm2d(O).cuda().eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(O, device='cuda', generator=g) + 0.5); bn.bias.copy_(rnd(O) * 0.1)
            bn.running_mean.copy_(rnd(O) * 0.1); bn.running_var.copy_(torch.rand(O, device='cuda', generator=g) + 0.5)
    else:
        bias = (rnd(O) * 0.1).requires_grad_()
    oh, ow = ho.out_hw(H, W, R, R, c['stride'], c['pad'], 1)
    res = rnd(B, O, oh, ow) if c['res'] else None
    # ---- HIP, X-layout
    xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
    rx = AF.as_nchw(_x(res), B, oh, ow) if res is not None else None
    y = AF.conv_bn_act(xx, w, bn=bn, bias=bias, res=rx, stride=c['stride'], pad=c['pad'], relu=c['relu'], out_f32=c['out_f32'])
    yf = y if c['out_f32'] else _f(AF.as_rows(y), B, oh, ow, O)
    # ---- torch fp32
    xr = x.clone().requires_grad_()
    wr = w.detach().clone().requires_grad_()
    z = F.conv2d(xr, wr, bias.detach() if bias is not None else None, c['stride'], c['pad'])
    if bn is not None:
        z = F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight.detach(), bn.bias.detach(), False, 0.0, bn.eps)
    if res is not None:
        z = z + res
    if c['relu']:
        # the ReLU of the HIP output decides the mask on both sides: an output within rounding distance of zero may have either sign, and
        # a flipped mask bit changes the gradients of a 3 x 3 neighbourhood by O(1) -- that is the activation, not the arithmetic under test
        z = z * (yf.detach() > 0).float()
    assert _err(yf, z) < 1e-4, ('forward', _err(yf, z))
    # ---- backward
    gy = rnd(B, O, oh, ow)
    if c['out_f32']:
        y.backward(gy)
    else:
        y.backward(AF.as_nchw(_x(gy), B, oh, ow))
    z.backward(gy)
    torch.cuda.synchronize()
    if C % 32 == 0:
        gx = _f(AF.as_rows(xx.grad), B, H, W, C)
        assert _err(gx, xr.grad) < 1e-4, ('dgrad', _err(gx, xr.grad))
    assert _err(w.grad, wr.grad) < 1e-4, ('wgrad', _err(w.grad, wr.grad))
    if bn is not None:
        # eval-mode BN parameter gradients through the fused path (<w, dW> and the column sums)
        zz = F.conv2d(x, w.detach(), None, c['stride'], c['pad'])
        assert bn.weight.grad is not None and bn.bias.grad is not None
        gm = gy * (yf.detach() > 0) if c['relu'] else gy
        ref_beta = gm.sum((0, 2, 3))
        ref_gamma = (gm * (zz - bn.running_mean[None, :, None, None]) * torch.rsqrt(bn.running_var + bn.eps)[None, :, None, None]).sum((0, 2, 3))
        assert _err(bn.bias.grad, ref_beta) < 1e-4 and _err(bn.weight.grad, ref_gamma) < 2e-4, (_err(bn.bias.grad, ref_beta), _err(bn.weight.grad, ref_gamma))
    else:
        gm = gy * (yf.detach() > 0) if c['relu'] else gy
        assert _err(bias.grad, gm.sum((0, 2, 3))) < 1e-4


def test_x3_row_kernels_against_fp32():
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.hipops import Seg
    g = torch.Generator(device='cuda').manual_seed(9)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    B, C, H, W = 2, 64, 18, 22
    x = rnd(B, C, H, W)
    # max-pool
    out, s = ho.maxpool3x3s2(_x(x), Seg(B, H, W))
    assert _err(_f(out, B, s.H, s.W, C), F.max_pool2d(x, 3, 2, 1)) < 2e-5
    # upsample-add and its adjoint
    top = rnd(B, C, H // 2, W // 2)
    up = ho.upsample_add(_x(x), Seg(B, H, W), _x(top), Seg(B, H // 2, W // 2))
    ref = x + F.interpolate(top, size=(H, W), mode='nearest')
    assert _err(_f(up, B, H, W, C), ref) < 2e-5
    gb = ho.upsample_add_bwd(_x(x), Seg(B, H, W), Seg(B, H // 2, W // 2))
    tr = top.clone().requires_grad_()
    (F.interpolate(tr, size=(H, W), mode='nearest') * x).sum().backward()
    assert _err(_f(gb, B, H // 2, W // 2, C), tr.grad) < 2e-5
    # activation backward + column sums
    a = torch.relu(rnd(B, C, H, W))
    dz, _, s1, _ = ho.act_bwd(_x(x), _x(a), relu=True)
    refdz = x * (a > 0)
    assert _err(_f(dz, B, H, W, C), refdz) < 2e-5 and _err(s1[:C], refdz.sum((0, 2, 3))) < 1e-5
    _, _, s1b, _ = ho.act_bwd(_x(x), None, relu=False, want_dz=False)
    assert _err(s1b[:C], x.sum((0, 2, 3))) < 1e-5
    # head-gradient cast (+ fused ReLU of retina_L) and column sums
    for N in (180, 36, 9):
        gg = rnd(500, N)
        ro = rnd(500, N) if N == 9 else None
        dzp, cs = ho.pad_cast_colsum(gg, ho.xw(N), ro)
        refg = gg * (ro > 0) if ro is not None else gg
        assert _err(ho.x3_merge(dzp, N), refg) < 2e-5 and _err(cs[:N], refg.sum(0)) < 1e-5
        if N % 32:
            assert float(ho.x3_merge(dzp)[:, N:].abs().max()) == 0.0
    # the 8-columns-per-thread form (N % 4 == 0) over many blocks, with the ReLU mask: heads / tails are the roundings of x3_split, bit for bit
    for N in (180, 36, 64):
        gg, ro = rnd(70001, N), rnd(70001, N)
        dzp, cs = ho.pad_cast_colsum(gg, ho.xw(N), ro)
        refg = gg * (ro > 0)
        assert torch.equal(dzp[:, :ho.xw(N)].contiguous(), ho.x3_split(torch.nn.functional.pad(refg, (0, ho.xw(N) // 2 - N))))
        assert _err(cs[:N], refg.sum(0)) < 1e-5
    # fan-in add
    y = rnd(B, C, H, W)
    assert _err(ho.x3_merge(ho.x3_add(_x(x), _x(y)), C), (x + y).permute(0, 2, 3, 1).reshape(-1, C)) < 2e-5
    # fork: two consumers, gradients added on the values
    xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
    p, q = AF.fork(xx, 2)
    g1, g2 = rnd(B, C, H, W), rnd(B, C, H, W)
    torch.autograd.backward([p, q], [AF.as_nchw(_x(g1), B, H, W), AF.as_nchw(_x(g2), B, H, W)])
    assert _err(_f(AF.as_rows(xx.grad), B, H, W, C), g1 + g2) < 2e-5


@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 70, 102), (3, 128, 192), (1, 34, 30)])
def test_x3_stem_against_fp32(shape, monkeypatch):
    """frozen stem in the reference-precision mode, both forms: the ONE-launch kernel (csrc/stem_x3.hip: fp32 image -> pooled X rows) and
    the three launches it replaces (space-to-depth X rows of 64 columns -> 4x4 conv + BN + ReLU -> max-pool), each against torch fp32 and
    against each other (they group the fp32 sums differently: equal to summation order, not to the bit); ragged tiles included"""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.mmcv_lite import BatchNorm2d, Conv2d
    g = torch.Generator(device='cuda').manual_seed(3)
    conv = Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
    bn = BatchNorm2d(64).cuda().eval()
    for q in list(conv.parameters()) + list(bn.parameters()):
        q.requires_grad_(False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(64, 3, 7, 7, device='cuda', generator=g) * 0.05)
        bn.running_mean.copy_(torch.randn(64, device='cuda', generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(64, device='cuda', generator=g) + 0.5)
    B, H, W = shape
    img = torch.randn(B, 3, H, W, device='cuda', generator=g)
    assert AF.stem_s2d_applies(img, conv, bn)
    ref = F.max_pool2d(torch.relu(F.batch_norm(F.conv2d(img, conv.weight, None, 2, 3), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)), 3, 2, 1)
    got = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('AOD_STEM_POOL_FUSE', fuse)
        y = AF.stem_pool_s2d(img, conv, bn)
        got[fuse] = AF.x3_to_f32(y, 64)
        assert got[fuse].shape == ref.shape and _err(got[fuse], ref) < 1e-4, (fuse, _err(got[fuse], ref))
    # (a value's tail has 8 bits below its head's 8: one flipped tail rounding is 2^-17 = 7.6e-6 of the value)
    assert _err(got['1'], got['0']) < 2e-5, _err(got['1'], got['0'])


def test_x3_grouped_tower_launches_equal_the_separate_ones():
    """cls / reg tower convs of one depth as ONE grouped launch of the 256 x 256 tile (aod_conv2d_grouped, x3), forward and dgrad with the fused
    activation backward, against two separate launches of the 4-wave tile: the same K order per output element -> identical bits"""
    import os
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    g = torch.Generator(device='cuda').manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    B, C = 2, 256
    shapes = [(24, 40), (12, 20), (6, 10)]
    convs = []
    for _ in range(4):
        c = Conv2d(C, C, 3, padding=1).cuda()
        with torch.no_grad():
            c.weight.copy_(rnd(C, C, 3, 3) * 0.02); c.bias.copy_(rnd(C) * 0.1)
        convs.append(c)
    _, slots = AF.pyramid_buffer([(B, h, w) for h, w in shapes], C, 'cuda')
    feats = []
    for sl, (h, w) in zip(slots, shapes):
        sl.copy_(AF.as_nchw(_x(rnd(B, C, h, w)), B, h, w))
        feats.append(sl)

    def run(grouped):
        os.environ['AOD_GROUP_TOWERS'] = '1' if grouped else '0'
        try:
            xs = [f.detach().requires_grad_() for f in feats]
            a, b = zip(*[AF.fork(x, 2) for x in xs])
            ya, yb = AF.conv_pair_act(list(a), list(b), convs[0], convs[1], sole_consumer=False)
            za, zb = AF.conv_pair_act(ya, yb, convs[2], convs[3], sole_consumer=True)
            for c in convs:
                c.weight.grad = c.bias.grad = None
            gs = [AF.as_nchw(_x(torch.randn(B, C, h, w, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5 + i))), B, h, w)
                  for i, (h, w) in enumerate(shapes)]
            torch.autograd.backward(list(za) + list(zb), gs + gs)
            torch.cuda.synchronize()
            return ([t.detach().clone() for t in list(za) + list(zb)], [x.grad.clone() for x in xs],
                    [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs])
        finally:
            os.environ.pop('AOD_GROUP_TOWERS', None)
    o1, gx1, gw1 = run(True)
    o0, gx0, gw0 = run(False)
    assert all(torch.equal(u, v) for u, v in zip(o1, o0))
    assert all(torch.equal(u, v) for u, v in zip(gx1, gx0))
    # weight / bias gradients: same slabs or a different grouping of the wgrad launches -> fp32 summation order may differ
    for u, v in zip(gw1, gw0):
        assert _err(u, v) < 1e-5
    # and against fp32
    x32 = [_f(AF.as_rows(f), B, h, w, C) for f, (h, w) in zip(feats, shapes)]
    ref = [torch.relu(F.conv2d(torch.relu(F.conv2d(x, convs[0].weight, convs[0].bias, 1, 1)), convs[2].weight, convs[2].bias, 1, 1)) for x in x32]
    for r, o, (h, w) in zip(ref, o1[:3], shapes):
        assert _err(_f(AF.as_rows(o), B, h, w, C), r) < 1e-4


def test_x3_ssd_row_kernels_against_fp32():
    """generic max-pool forward / backward (ceil-mode 2x2 s2 and 3x3 s1 p1, first-maximum routing like torch) and L2Norm forward / backward on X rows"""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import functional_ssd as FS
    g = torch.Generator(device='cuda').manual_seed(21)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    B, C, H, W = 2, 64, 19, 23
    x = rnd(B, C, H, W)
    for k, s, p, ceil in ((2, 2, 0, True), (3, 1, 1, False)):
        xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
        y = FS.max_pool(xx, k, s, p, ceil)
        xr = x.clone().requires_grad_()
        ref = F.max_pool2d(xr, k, s, p, ceil_mode=ceil)
        oh, ow = ref.shape[2:]
        assert tuple(y.shape[2:]) == (oh, ow) and _err(_f(AF.as_rows(y), B, oh, ow, C), ref) < 2e-5
        gy = rnd(B, C, oh, ow)
        y.backward(AF.as_nchw(_x(gy), B, oh, ow))
        ref.backward(gy)
        assert _err(_f(AF.as_rows(xx.grad), B, H, W, C), xr.grad) < 2e-5
    w = (torch.rand(C, device='cuda', generator=g) * 20).requires_grad_()
    xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
    y = FS.l2norm(xx, w, 1e-10)
    xr, wr = x.clone().requires_grad_(), w.detach().clone().requires_grad_()
    ref = wr[None, :, None, None] * xr / (xr.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
    assert _err(_f(AF.as_rows(y), B, H, W, C), ref) < 2e-5
    gy = rnd(B, C, H, W)
    y.backward(AF.as_nchw(_x(gy), B, H, W))
    ref.backward(gy)
    assert _err(_f(AF.as_rows(xx.grad), B, H, W, C), xr.grad) < 5e-5 and _err(w.grad, wr.grad) < 5e-5


@pytest.mark.parametrize('shape', [(2, 32, 48), (1, 13, 37), (3, 8, 16)])
def test_x3_fused_bottleneck64_equals_the_three_launch_block(shape):
    """aod_bottleneck64x3_fwd (frozen layer-1 blocks in the reference-precision mode: conv1 on the halo, conv2 from LDS, conv3 + residual, one
    launch) against the same block as three x3 conv launches -- same products in the same order per accumulator -> identical bits -- and
    against fp32; whole and ragged tiles, the first block (64 input channels + downsample branch) and an identity block (256)"""
    import os
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.models.backbones.resnet import Bottleneck
    from aod_meh_hua_amd.mmcv_lite import BatchNorm2d, Conv2d
    import torch.nn as nn
    B, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(17)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    for cin in (64, 256):
        ds = None
        if cin != 256:
            ds = nn.Sequential(Conv2d(cin, 256, 1, bias=False), BatchNorm2d(256))
        blk = Bottleneck(cin, 64, downsample=ds).cuda().eval()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.copy_(rnd(*m.weight.shape) / (m.weight[0].numel()) ** 0.5)
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(torch.rand(m.weight.shape, device='cuda', generator=g) + 0.5); m.bias.copy_(rnd(*m.bias.shape) * 0.1)
                    m.running_mean.copy_(rnd(*m.bias.shape) * 0.1); m.running_var.copy_(torch.rand(m.bias.shape, device='cuda', generator=g) + 0.5)
        for q in blk.parameters():
            q.requires_grad_(False)
        x = rnd(B, cin, H, W)
        xx = AF.as_nchw(_x(x), B, H, W)
        with torch.no_grad():
            assert AF.bottleneck64_applies(blk, xx)
            y1 = blk(xx)
            os.environ['AOD_FUSE_BOTTLENECK_X3'] = '0'
            try:
                assert not AF.bottleneck64_applies(blk, xx)
                y0 = blk(xx)
            finally:
                os.environ.pop('AOD_FUSE_BOTTLENECK_X3', None)
            torch.cuda.synchronize()
            bnf = lambda z, n: F.batch_norm(z, n.running_mean, n.running_var, n.weight, n.bias, False, 0.0, n.eps)
            t = torch.relu(bnf(F.conv2d(x, blk.conv1.weight), blk.norm1))
            t = torch.relu(bnf(F.conv2d(t, blk.conv2.weight, None, 1, 1), blk.norm2))
            idn = x if ds is None else bnf(F.conv2d(x, ds[0].weight), ds[1])
            ref = torch.relu(bnf(F.conv2d(t, blk.conv3.weight), blk.norm3) + idn)
        f1 = _f(AF.as_rows(y1), B, H, W, 256)
        assert _err(f1, ref) < 1e-4, (cin, _err(f1, ref))
        assert torch.equal(y1, y0), (cin, float((AF.as_rows(y1).float() - AF.as_rows(y0).float()).abs().max()))
        # phase 1 on two LDS stages instead of three (AOD_B64X3_ST3=0): the same K order, the same bits
        os.environ['AOD_B64X3_ST3'] = '0'
        try:
            with torch.no_grad():
                y3 = blk(xx)
        finally:
            os.environ.pop('AOD_B64X3_ST3', None)
        assert torch.equal(y1, y3)
        if ds is not None:
            # the first block's downsample branch rides in the launch (aod_bottleneck64x3_ds_fwd); as a launch of its own: the same bits again
            assert AF.bottleneck64_ds_fused(blk, xx)
            os.environ['AOD_FUSE_BOTTLENECK_DS'] = '0'
            try:
                with torch.no_grad():
                    assert not AF.bottleneck64_ds_fused(blk, xx) and AF.bottleneck64_applies(blk, xx)
                    y2 = blk(xx)
            finally:
                os.environ.pop('AOD_FUSE_BOTTLENECK_DS', None)
            assert torch.equal(y1, y2)


@pytest.mark.parametrize('shape', [(2, 16, 32), (1, 13, 37), (3, 7, 129), (16, 64, 64)])
def test_x3_fused_bottleneck128_equals_the_three_launch_block(shape):
    """aod_bottleneck128x3_fwd (identity blocks of the 128-plane stage in the reference-precision mode: conv1 on the halo, conv2 with the tap
    innermost against a streamed filter ring, conv3 + residual in registers, one launch) against the same block as three x3 conv launches --
    same products in the same order per accumulator -> identical bits --, as the inference / frozen forward and as the forward of a training
    step (kept intermediates, autograd nodes recorded around the launch's outputs: identical gradients), and against fp32; whole and ragged
    tiles."""
    import os
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.models.backbones.resnet import Bottleneck
    import torch.nn as nn
    B, H, W = shape
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
    x = rnd(B, 512, H, W)
    xx = AF.as_nchw(_x(x), B, H, W)

    def run(fused, train):
        os.environ['AOD_FUSE_BOTTLENECK128_X3'] = '1' if fused else '0'
        try:
            if not train:
                with torch.no_grad():
                    assert AF.bottleneck128_applies(blk, xx) == fused
                    return blk(xx), None
            xi = xx.detach().clone().requires_grad_()
            assert AF.bottleneck128_train_applies(blk, xi) == fused
            for q in blk.parameters():
                q.grad = None
            y = blk(xi)
            gy = AF.as_nchw(_x(torch.randn(B, 512, H, W, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))), B, H, W)
            y.backward(gy)
            torch.cuda.synchronize()
            return y.detach(), [xi.grad.clone()] + [q.grad.clone() for q in blk.parameters()]
        finally:
            os.environ.pop('AOD_FUSE_BOTTLENECK128_X3', None)

    y1, _ = run(True, False)
    y0, _ = run(False, False)
    torch.cuda.synchronize()
    with torch.no_grad():
        bnf = lambda z, n: F.batch_norm(z, n.running_mean, n.running_var, n.weight, n.bias, False, 0.0, n.eps)
        t = torch.relu(bnf(F.conv2d(x, blk.conv1.weight), blk.norm1))
        t = torch.relu(bnf(F.conv2d(t, blk.conv2.weight, None, 1, 1), blk.norm2))
        ref = torch.relu(bnf(F.conv2d(t, blk.conv3.weight), blk.norm3) + x)
    f1 = _f(AF.as_rows(y1), B, H, W, 512)
    assert _err(f1, ref) < 1e-4, _err(f1, ref)
    if not torch.equal(y1, y0):
        d = (AF.as_rows(y1).float() - AF.as_rows(y0).float()).abs()
        idx = d.nonzero()
        raise AssertionError(f'{int(idx.shape[0])} elements differ, max {float(d.max()):.3e}; rows {idx[:, 0].unique().tolist()[:12]} '
                             f'(pixels {[(int(r) // W % H, int(r) % W) for r in idx[:, 0].unique()[:12]]}), columns {idx[:, 1].unique().tolist()[:12]}')
    # training forward: the same output bits, and -- because the kept intermediates are the bits the separate launches store -- the same gradients
    for q in blk.parameters():
        q.requires_grad_(True)
    yt1, g1 = run(True, True)
    yt0, g0 = run(False, True)
    assert torch.equal(yt1, y1) and torch.equal(yt0, y0)
    for a, b_ in zip(g1, g0):
        if a.dim() == 4:          # the input gradient and the filter gradients: identical bits
            assert torch.equal(a, b_), float((a.float() - b_.float()).abs().max())
        else:                     # BN gradients come from column sums (fp32 atomics: arrival order) in either form
            assert float((a - b_).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-7, float((a - b_).abs().max())


@pytest.mark.parametrize('shape', [(2, 21, 37), (3, 64, 64)])
def test_x3_fused_bottleneck128_backward_equals_the_three_dgrad_launches(shape, monkeypatch):
    """aod_bottleneck128x3_bwd (dgrad chain of an identity block of the 128-plane stage on X rows: the three products with the mask /
    skip-gradient / column-sum epilogues of the x3 dgrad launches, intermediate gradients in LDS) against the three launches, through autograd
    over a whole stage (downsample block + identity blocks: the ActSlot / skip-gradient hand-overs at both ends of every chain are exercised):
    input gradient and filter gradients identical bits, BN gradients (fp32 atomics in both forms) within 1e-5 of their scale."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.models.backbones.resnet import Bottleneck, ResLayer
    import torch.nn as nn
    B, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(29)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    layer = ResLayer(Bottleneck, 256, 128, 4, stride=2).cuda().eval()
    with torch.no_grad():
        for m in layer.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(rnd(*m.weight.shape) / (m.weight[0].numel()) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, device='cuda', generator=g) * 0.5 + 0.5); m.bias.copy_(rnd(*m.bias.shape) * 0.1)
                m.running_mean.copy_(rnd(*m.bias.shape) * 0.1); m.running_var.copy_(torch.rand(m.bias.shape, device='cuda', generator=g) + 0.5)
    x0 = AF.as_nchw(_x(rnd(B, 256, 2 * H, 2 * W).relu() * 0.5), B, 2 * H, 2 * W)
    gy = AF.as_nchw(_x(rnd(B, 512, H, W) * 0.1), B, H, W)
    params = [q for q in layer.parameters()]
    res, calls = {}, {}
    orig = ho.bottleneck_bwd
    for mode in ('0', '1'):
        monkeypatch.setenv('AOD_FUSE_BOTTLENECK128_X3_BWD', mode)
        n = [0]
        monkeypatch.setattr(ho, 'bottleneck_bwd', lambda *a, **k: (n.__setitem__(0, n[0] + 1), orig(*a, **k))[1])
        x = x0.detach().clone().requires_grad_(True)
        y = layer(x)
        grads = torch.autograd.grad(y, [x] + params, gy)
        torch.cuda.synchronize()
        res[mode], calls[mode] = [t.clone() for t in grads], n[0]
    assert calls['0'] == 0 and calls['1'] == len(layer) - 1          # every identity block took the fused chain
    names = ['x'] + [n_ for n_, q in layer.named_parameters()]
    for n_, a, b_ in zip(names, res['0'], res['1']):
        if a.dim() == 4:
            assert torch.equal(a, b_), (n_, float((a.float() - b_.float()).abs().max()))
        else:
            scale = float(a.abs().max()) + 1e-12
            assert float((a - b_).abs().max()) <= 1e-5 * scale + 1e-7, (n_, float((a - b_).abs().max()), scale)
    assert float(res['1'][0].float().abs().mean()) > 0


@pytest.mark.parametrize('N,relu', [(36, False), (9, True), (48, False), (20, True)])
def test_x3_narrow_prediction_convs_on_the_halo_kernel_equal_the_general_kernel(N, relu, monkeypatch):
    """aod_halo_conv3x3_x3 (csrc/halo_x3.hip: retina_reg / retina_L of Lambda_L2.py:52-54,100-103 in the reference-precision mode -- a 32-channel
    chunk of the 10 x 18 halo and of all nine taps' filter slices resident in LDS per K-step) against the general implicit-GEMM kernel, which walks
    these layers in the same (chunk, tap) order: identical bits over a five-level pyramid with whole, ragged and single-pixel levels; and against
    fp32."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    g = torch.Generator(device='cuda').manual_seed(31 + N)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    conv = Conv2d(256, N, 3, padding=1).cuda()
    with torch.no_grad():
        conv.weight.copy_(rnd(N, 256, 3, 3) / 48.0); conv.bias.copy_(rnd(N) * 0.1)
    for B, sizes in ((2, ((16, 16), (8, 8), (4, 4), (2, 2), (1, 1))), (3, ((21, 37), (11, 19), (6, 10), (3, 5), (2, 3))), (16, ((64, 64),))):
        xs = [rnd(B, 256, h, w) for h, w in sizes]
        feats = [AF.as_nchw(_x(x), B, x.shape[2], x.shape[3]) for x in xs]
        calls = []
        orig = ho.call
        monkeypatch.setattr(ho, 'call', lambda name, *a: (calls.append(name), orig(name, *a))[1])
        outs = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('AOD_HALO_X3', mode)
            monkeypatch.setattr(ho, 'SPLITK', False)        # (small pyramids: the general kernel would slice K, a different summation order)
            calls.clear()
            with torch.no_grad():
                outs[mode] = [o.clone() for o in conv(list(feats), out_f32=True, relu=relu)]
            assert ('aod_halo_conv3x3_x3' in calls) == (mode == '1'), calls
        # the two-rows-per-wave form of the kernel (AOD_HALO_X3_RW=2, four waves): the same bits again
        monkeypatch.setenv('AOD_HALO_X3', '1'); monkeypatch.setenv('AOD_HALO_X3_RW', '2')
        with torch.no_grad():
            outs['rw2'] = [o.clone() for o in conv(list(feats), out_f32=True, relu=relu)]
        monkeypatch.undo()
        torch.cuda.synchronize()
        assert all(torch.equal(a, c) for a, c in zip(outs['1'], outs['rw2']))
        for x, a, b_ in zip(xs, outs['1'], outs['0']):
            ref = F.conv2d(x, conv.weight, conv.bias, 1, 1)
            ref = torch.relu(ref) if relu else ref
            assert a.shape == ref.shape and _err(a, ref) < 1e-4, _err(a, ref)
            assert torch.equal(a, b_), float((a - b_).abs().max())


def test_x3_retina_cls_on_the_192_tile_equals_the_64_wide_tiles(monkeypatch):
    """retina_cls (Lambda_L2.py:52, 9 anchors x 20 classes = 180 fp32 columns) at configs[1]'s pyramid (16 x 512^2: 87 296 rows) takes the
    192 x 192 eight-wave tile (conv.hip, AOD_X3_TILE_192); same K order and products as the 128 x 64 tile it replaces -> identical bits; a
    sample of rows against fp32."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    g = torch.Generator(device='cuda').manual_seed(77)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    conv = Conv2d(256, 180, 3, padding=1).cuda()
    with torch.no_grad():
        conv.weight.copy_(rnd(180, 256, 3, 3) / 48.0); conv.bias.copy_(rnd(180) * 0.1)
    B = 16
    xs = [rnd(B, 256, s, s) for s in (64, 32, 16, 8, 4)]
    feats = [AF.as_nchw(_x(x), B, x.shape[2], x.shape[3]) for x in xs]
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('AOD_X3_TILE_192', mode)
        with torch.no_grad():
            outs[mode] = [o.clone() for o in conv(list(feats), out_f32=True)]
    torch.cuda.synchronize()
    for x, a, b_ in zip(xs, outs['1'], outs['0']):
        assert torch.equal(a, b_), float((a - b_).abs().max())
        ref = F.conv2d(x[:2], conv.weight, conv.bias, 1, 1)
        assert _err(a[:2], ref) < 1e-4, _err(a[:2], ref)
