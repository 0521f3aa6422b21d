"""GPU: the persistent producer / consumer x3 convolution (csrc/conv_x3p.hip; the backbone's and the neck's 1x1 / 3x3 layers,
mmdet/models/backbones/resnet.py:262-301, necks/fpn.py:151-202) against the general implicit-GEMM kernel it replaces: the same products in the
same order and the same fp32 epilogue -> IDENTICAL BITS for the outputs and the input gradients (forward with BN / bias / residual / ReLU at
stride 1 and 2, dgrad with ReLU mask / deferred residual gradient), whole, ragged and multi-round tile counts, one and several segments; the
fused column sums (bias / BN-shift gradients) are fp32 atomics in both kernels and agree to rounding; and against torch fp32."""
import os

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


def _x(t):
    from aod_meh_hua_amd import hipops as ho
    B, C, H, W = t.shape
    return ho.x3_split(t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())


def _f(rows, B, H, W, C):
    from aod_meh_hua_amd import hipops as ho
    return ho.x3_merge(rows, C).view(B, H, W, C).permute(0, 3, 1, 2)


def _err(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))


CASES = [
    # B, C, O, H, W, R, stride, bn, res, relu      (tiles of 128 x 128: rows / 128 x O / 128)
    dict(B=2, C=256, O=128, H=32, W=32, R=1, stride=1, bn=True, res=False, relu=True),        # 16 tiles, (tap, chunk) order irrelevant (1x1)
    dict(B=2, C=128, O=128, H=32, W=32, R=3, stride=1, bn=True, res=False, relu=True),        # 3x3, C < 256: taps outermost
    dict(B=2, C=256, O=256, H=32, W=32, R=3, stride=1, bn=True, res=False, relu=True),        # 3x3, C >= 256: taps innermost (layer-3 conv2)
    dict(B=3, C=128, O=512, H=24, W=40, R=1, stride=1, bn=True, res=True, relu=True),         # expand conv with residual = head + tail; ragged last tile (2 880 rows)
    dict(B=4, C=256, O=128, H=64, W=64, R=3, stride=2, bn=True, res=False, relu=True),        # stride-2 forward (first block's conv2): x3p forward, general dgrad
    dict(B=2, C=512, O=256, H=32, W=32, R=1, stride=2, bn=True, res=False, relu=False),       # 1x1 / stride-2 downsample
    dict(B=16, C=256, O=1024, H=32, W=32, R=1, stride=1, bn=True, res=True, relu=True),       # 1 024 tiles: four rounds of the persistent grid
    dict(B=1, C=64, O=128, H=13, W=21, R=3, stride=1, bn=False, res=False, relu=False),       # one K-chunk per tap, 273 rows: three tiles, the last ragged; bias
    dict(B=3, C=128, O=256, H=20, W=28, R=3, stride=2, bn=True, res=False, relu=True),        # class-major dgrad with ragged class tiles (3 x 10 x 14 = 420 rows per class)
    dict(B=2, C=256, O=256, H=32, W=32, R=3, stride=2, bn=False, res=False, relu=False),      # ... with 256-column tiles available
]


@pytest.mark.parametrize('case', CASES)
def test_x3p_forward_and_dgrad_equal_the_general_kernel(case, monkeypatch):
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import lib
    from aod_meh_hua_amd.mmcv_lite import BatchNorm2d
    c = case
    B, C, O, H, W, R, st = c['B'], c['C'], c['O'], c['H'], c['W'], c['R'], c['stride']
    pad = R // 2
    g = torch.Generator(device='cuda').manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    x = rnd(B, C, H, W)
    w0 = rnd(O, C, R, R) / (C * R * R) ** 0.5
    bn, bias0 = None, None
    if c['bn']:
        bn = BatchNorm2d(O).cuda().eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(O, device='cuda', generator=g) + 0.5); bn.bias.copy_(rnd(O) * 0.1)
            bn.running_mean.copy_(rnd(O) * 0.1); bn.running_var.copy_(torch.rand(O, device='cuda', generator=g) + 0.5)
    else:
        bias0 = rnd(O) * 0.1
    oh, ow = ho.out_hw(H, W, R, R, st, pad, 1)
    res = rnd(B, O, oh, ow) if c['res'] else None
    gy = rnd(B, O, oh, ow)
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P_MIN_STEPS', '1')      # (the product sends only tiles of >= 24 K-steps here; the kernel itself takes any)
    monkeypatch.setattr(ho, 'SPLITK', False)
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('AOD_X3P', mode)
        n0 = lib.aod_conv_x3p_count()
        w = w0.clone().requires_grad_()
        bias = bias0.clone().requires_grad_() if bias0 is not None else None
        if bn is not None:
            bn.weight.grad = bn.bias.grad = None
        xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
        rx = AF.as_nchw(_x(res), B, oh, ow) if res is not None else None
        y = AF.conv_bn_act(xx, w, bn=bn, bias=bias, res=rx, stride=st, pad=pad, relu=c['relu'])
        y.backward(AF.as_nchw(_x(gy), B, oh, ow))
        torch.cuda.synchronize()
        took = lib.aod_conv_x3p_count() - n0
        # forward always qualifies; the dgrad with C % 128 == 0 at stride 1, and at stride 2 for a 3x3 conv on an even map (the class-major
        # lattice form); a stand-alone 1x1 / stride-2 dgrad and 64-channel destinations stay with the general kernel
        dgrad_too = C % 128 == 0 and (st == 1 or (R == 3 and H % 2 == 0 and W % 2 == 0))
        assert took == (0 if mode == '0' else (2 if dgrad_too else 1)), (mode, took)
        out[mode] = dict(y=AF.as_rows(y).detach().clone(), gx=AF.as_rows(xx.grad).clone(), gw=w.grad.clone(),
                         gb=(bn.bias.grad if bn is not None else bias.grad).clone())
    a, b_ = out['1'], out['0']
    assert torch.equal(a['y'], b_['y']), float((a['y'].float() - b_['y'].float()).abs().max())
    assert torch.equal(a['gx'], b_['gx']), float((a['gx'].float() - b_['gx'].float()).abs().max())
    assert torch.equal(a['gw'], b_['gw'])                       # (same wgrad kernel, fed bit-identical gradients)
    assert _err(a['gb'], b_['gb']) < 2e-6                       # column sums: fp32 atomics, arrival order
    # ---- and against torch fp32
    z = F.conv2d(x, w0, bias0, st, pad)
    if bn is not None:
        z = F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight.detach(), bn.bias.detach(), False, 0.0, bn.eps)
    if res is not None:
        z = z + res
    yf = _f(a['y'], B, oh, ow, O)
    if c['relu']:
        z = z * (yf > 0).float()
    assert _err(yf, z) < 1e-4, _err(yf, z)


def test_x3p_pyramid_segments_and_mask_epilogue(monkeypatch):
    """Several segments sharing a filter (the neck's / towers' level-batched launches, Lambda_L2.py:85-94 style) whose row counts are multiples
    of 128 -- a tile never straddles two levels -- take the persistent kernel as ONE launch; a pyramid with a ragged inner level must not.  The
    dgrad epilogue with a ReLU mask and column sums (functional.ActSlot protocol: conv -> ReLU -> conv, the second conv's dgrad masks) is
    covered by chaining two convs."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import lib
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    g = torch.Generator(device='cuda').manual_seed(12)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P_MIN_STEPS', '1')
    monkeypatch.setattr(ho, 'SPLITK', False)
    c1, c2 = Conv2d(256, 256, 3, padding=1).cuda(), Conv2d(256, 128, 3, padding=1).cuda()
    with torch.no_grad():
        c1.weight.copy_(rnd(256, 256, 3, 3) / 48.0); c1.bias.copy_(rnd(256) * 0.1)
        c2.weight.copy_(rnd(128, 256, 3, 3) / 48.0); c2.bias.copy_(rnd(128) * 0.1)
    for B, sizes, expect in ((2, ((32, 32), (16, 16), (8, 8)), True), (2, ((32, 32), (12, 12), (8, 8)), False)):
        xs = [rnd(B, 256, h, w) for h, w in sizes]
        gys = [rnd(B, 128, h, w) for h, w in sizes]
        res = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('AOD_X3P', mode)
            n0 = lib.aod_conv_x3p_count()
            for m in (c1, c2):
                m.weight.grad = m.bias.grad = None
            feats = [AF.as_nchw(_x(x), B, x.shape[2], x.shape[3]).requires_grad_() for x in xs]
            hs = c1(list(feats), relu=True)
            ys = c2(list(hs), relu=False)
            torch.autograd.backward(list(ys), [AF.as_nchw(_x(gq), B, gq.shape[2], gq.shape[3]) for gq in gys])
            torch.cuda.synchronize()
            took = lib.aod_conv_x3p_count() - n0
            assert (took > 0) == (mode == '1' and expect), (mode, expect, took)
            res[mode] = ([AF.as_rows(y).detach().clone() for y in ys], [AF.as_rows(f.grad).clone() for f in feats],
                         c1.bias.grad.clone(), c2.bias.grad.clone(), c1.weight.grad.clone())
        for a, b_ in zip(res['1'][0] + res['1'][1], res['0'][0] + res['0'][1]):
            assert torch.equal(a, b_), float((a.float() - b_.float()).abs().max())
        assert _err(res['1'][2], res['0'][2]) < 2e-6 and _err(res['1'][3], res['0'][3]) < 2e-6
        assert _err(res['1'][4], res['0'][4]) < 2e-6
        ref = F.conv2d(torch.relu(F.conv2d(xs[0], c1.weight, c1.bias, 1, 1)), c2.weight, c2.bias, 1, 1)
        assert _err(_f(res['1'][0][0], B, sizes[0][0], sizes[0][1], 128), ref) < 2e-4


@pytest.mark.parametrize('shape', [(16, 256, 1024, 32, 32), (3, 128, 512, 24, 40), (5, 512, 256, 17, 23)])
def test_x3p_operand_prefetch_forms_equal_the_general_kernel(shape, monkeypatch):
    """1x1 launches with exactly one epilogue operand take the persistent kernel's PRE instances (the operand rows are requested before the
    tile's K loop): PRE = 1, the residual of a bottleneck's expand conv (resnet.py:283-298), and PRE = 2, the ReLU mask in the dgrad of a 1x1
    conv whose input is a ReLU output (functional.ActSlot: conv -> ReLU -> 1x1 conv).  Bits equal to the general kernel and to the persistent
    kernel without the prefetch; repeated launches stay bit-stable (the waits on the operand are counted by the compiler, the ring's by hand)."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import lib
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    B, C, O, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(21)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P_MIN_STEPS', '1')
    monkeypatch.setenv('AOD_X3P_PRE_MIN_STEPS', '1')
    monkeypatch.setattr(ho, 'SPLITK', False)
    c1, c2 = Conv2d(C, C, 1).cuda(), Conv2d(C, O, 1).cuda()
    with torch.no_grad():
        c1.weight.copy_(rnd(C, C, 1, 1) / C ** 0.5); c1.bias.copy_(rnd(C) * 0.1)
        c2.weight.copy_(rnd(O, C, 1, 1) / C ** 0.5); c2.bias.copy_(rnd(O) * 0.1)
    x, res, gy = rnd(B, C, H, W), rnd(B, O, H, W), rnd(B, O, H, W)
    outs = {}
    for mode in ('pre', 'nopre', 'general'):
        monkeypatch.setenv('AOD_X3P', '0' if mode == 'general' else '1')
        monkeypatch.setenv('AOD_X3P_PRE', '0' if mode == 'nopre' else '1')
        runs = []
        for rep in range(3 if mode == 'pre' else 1):
            for m in (c1, c2):
                m.weight.grad = m.bias.grad = None
            n0 = lib.aod_conv_x3p_count()
            xx = AF.as_nchw(_x(x), B, H, W).requires_grad_()
            h = c1(xx, relu=True)                                            # producer: its ReLU output feeds c2 only
            y = AF.conv_bn_act(h, c2.weight, bias=c2.bias, res=AF.as_nchw(_x(res), B, H, W), relu=True, sole_consumer=True)   # PRE = 1 forward, PRE = 2 dgrad
            y.backward(AF.as_nchw(_x(gy), B, H, W))
            torch.cuda.synchronize()
            took = lib.aod_conv_x3p_count() - n0
            assert (took > 0) == (mode != 'general'), (mode, took)
            runs.append((AF.as_rows(y).detach().clone(), AF.as_rows(xx.grad).clone(), c1.weight.grad.clone(), c1.bias.grad.clone()))
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])
        outs[mode] = runs[0]
    for other in ('nopre', 'general'):
        assert torch.equal(outs['pre'][0], outs[other][0]), (other, 'forward')
        assert torch.equal(outs['pre'][1], outs[other][1]), (other, 'input gradient')
        assert torch.equal(outs['pre'][2], outs[other][2]), (other, 'weight gradient of the producer')
        assert _err(outs['pre'][3], outs[other][3]) < 2e-6
    ref = F.conv2d(torch.relu(F.conv2d(x, c1.weight, c1.bias)), c2.weight, c2.bias) + res
    yf = _f(outs['pre'][0], B, H, W, O)
    assert _err(yf, ref * (yf > 0).float()) < 2e-4


def test_x3p_repeated_launches_are_bit_stable(monkeypatch):
    """race screen: the ring's waits are counted by hand -- 40 launches of a multi-round shape, back to back, must all give the first one's bits"""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    g = torch.Generator(device='cuda').manual_seed(13)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P', '1')
    B, C, O, H, W = 16, 256, 256, 32, 32
    x = AF.as_nchw(_x(rnd(B, C, H, W)), B, H, W)
    w = rnd(O, C, 3, 3) / 48.0
    bias = rnd(O) * 0.1
    with torch.no_grad():
        first = AF.as_rows(AF.conv_bn_act(x, w, bias=bias, stride=1, pad=1, relu=True)).clone()
        junk = torch.empty(64 << 20, device='cuda', dtype=torch.uint8)
        for i in range(40):
            if i % 4 == 0:
                junk.random_(0, 255)                    # (evict the operands now and then: cold and warm loads take different times)
            y = AF.as_rows(AF.conv_bn_act(x, w, bias=bias, stride=1, pad=1, relu=True))
            assert torch.equal(y, first), i


@pytest.mark.parametrize('bn_cols', ['256', '128'])
def test_x3p_grouped_tower_launches_equal_the_256_tile(bn_cols, monkeypatch):
    """The head towers' grouped launches (cls / reg convs of one depth in one grid, Lambda_L2.py:44-51,85-94; aod_conv2d_grouped) on the
    persistent kernel -- 128 x 256 tiles, the tile index names the group -- against the 256 x 256 eight-wave tile they ran on: forward with bias +
    ReLU, grouped dgrad with the producers' ReLU masks and column sums; a pyramid whose levels are multiples of 128 rows.  Identical bits (the
    column sums to rounding), in both column-tile widths of the kernel."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import lib
    from aod_meh_hua_amd.mmcv_lite import Conv2d
    g = torch.Generator(device='cuda').manual_seed(21)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    B, C = 4, 256
    shapes = [(32, 32), (16, 16), (8, 8), (4, 8), (3, 5)]                 # 4 096 + 1 024 + 256 + 128 rows, and a ragged last level (60 rows)
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
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P_BN', bn_cols)
    monkeypatch.setenv('AOD_GROUP_TOWERS', '1')
    monkeypatch.setenv('AOD_X3P_GROUPED', '1')         # (opt-in in the product: level with the 256 x 256 tile, see conv.hip)
    monkeypatch.setattr(ho, 'SPLITK', False)

    def run(mode):
        monkeypatch.setenv('AOD_X3P', mode)
        n0 = lib.aod_conv_x3p_count()
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
        took = lib.aod_conv_x3p_count() - n0
        return ([t.detach().clone() for t in list(za) + list(zb)], [x.grad.clone() for x in xs],
                [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs], took)
    o1, gx1, gw1, took1 = run('1')
    o0, gx0, gw0, took0 = run('0')
    assert took0 == 0 and took1 >= 3, (took0, took1)          # two grouped forwards + at least the grouped dgrad of the second depth
    assert all(torch.equal(u, v) for u, v in zip(o1, o0))
    assert all(torch.equal(u, v) for u, v in zip(gx1, gx0))
    for u, v in zip(gw1, gw0):
        assert _err(u, v) < 2e-6


def test_x3p_whole_training_step_equals_the_general_kernels(monkeypatch):
    """Everything the persistent kernel takes inside the model -- forward and dgrad of the backbone's / neck's 3x3 and deep 1x1 layers, the
    class-major stride-2 dgrads of the stages' first blocks, the IN-PLACE 1x1 / stride-2 dgrads of the gradient junctions (lattice launches) --
    against the general kernels in one training iteration: identical loss, identical weight gradients wherever no fp32-atomic column sum enters
    (conv weights of layers without BN / bias gradients upstream are bit-equal; everything else to 2e-6)."""
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import lib
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    from oracle import model as omodel
    from tests import synth
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    sd = omodel.seeded_state_dict()
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    B, H, W = 4, 256, 256
    gtb, gtl = synth.random_gts(B, H, W, seed=24, gmin=1, gmax=3)
    data = dict(img=synth.images(B, H, W).cuda(), img_metas=synth.metas(B, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    monkeypatch.setenv('AOD_X3P_MIN_TILES', '1')
    monkeypatch.setenv('AOD_X3P_MIN_STEPS', '1')
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('AOD_X3P', mode)
        n0 = lib.aod_conv_x3p_count()
        model.zero_grad()
        o, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        o['loss'].backward()
        torch.cuda.synchronize()
        took = lib.aod_conv_x3p_count() - n0
        assert (took > 40) == (mode == '1'), (mode, took)
        out[mode] = (float(o['loss']), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    assert out['1'][0] == out['0'][0], (out['1'][0], out['0'][0])
    worst = 0.0
    for k, g1 in out['1'][1].items():
        g0 = out['0'][1][k]
        worst = max(worst, _err(g1, g0))
    assert worst < 2e-6, worst
