"""GPU end-to-end parity: the product model (HIP kernels, bf16 MFMA convs) against the golden values the
REFERENCE produced for the same seeded weights/inputs (tests/golden/train_step.npz, fp32 CPU) and against
the fp32 oracle's gradients.  Tolerances reflect bf16 operands / fp32 accumulation through ~60 conv layers."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    sd = omodel.seeded_state_dict()
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    model.train()
    return model, sd


def test_state_dict_keys_match_reference(built):
    model, _ = built
    g = np.load(os.path.join(G, 'state_dict_spec.npz'))
    assert list(model.state_dict().keys()) == list(g['keys'])
    assert [str(tuple(v.shape)) for v in model.state_dict().values()] == list(g['shapes'])
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(g['n_trainable'])


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_train_step_vs_reference_golden(built):
    model, sd = built
    g = np.load(os.path.join(G, 'train_step.npz'))
    H = W = 128
    img = synth.images(2, H, W).cuda()
    gtb, gtl = synth.random_gts(2, H, W, seed=24, gmin=1, gmax=3)
    data = dict(img=img, img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    torch.cuda.synchronize()
    assert int(head_out[8]) == int(g['num_total_samples'])                      # integer-exact assignment
    from aod_meh_hua_amd import functional as AF
    f32 = lambda t: AF.x3_to_f32(t)          # (X-layout rows in the reference-precision mode, the suite's default; identity in the fast mode)
    fam = [float(f32(f).abs().mean()) for f in feat_out]
    assert np.allclose(fam, g['feat_absmean'], rtol=2e-2), (fam, g['feat_absmean'])
    assert rel(f32(feat_out[4]).cpu().numpy(), g['feat_l4']) < 3e-2
    assert rel(head_out[1][3].detach().float().cpu().numpy(), g['cls_l3']) < 3e-2
    lv = [float(out['log_vars'][k]) for k in ('loss_cls', 'loss_bbox', 'loss_noR')]
    assert np.allclose(lv, g['log_vars'], rtol=2e-2), (lv, g['log_vars'])
    assert np.allclose(float(out['loss']), g['loss'], rtol=2e-2)
    assert rel(prev[4].cpu().numpy(), g['loss_noR_l4']) < 3e-2
    model.zero_grad()
    out['loss'].backward()
    torch.cuda.synchronize()
    pd = dict(model.named_parameters())
    gn = np.array([float(pd[k].grad.float().norm()) for k in g['grad_names']])
    assert np.allclose(gn, g['grad_norms'], rtol=5e-2), (gn, g['grad_norms'])
    assert pd['bbox_head.retina_L.weight'].grad is None and pd['backbone.conv1.weight'].grad is None
    lossL = model.train_step_L(prev, head_out, feat_out)
    model.zero_grad()
    lossL['loss'].backward()
    torch.cuda.synchronize()
    assert np.allclose(float(lossL['loss']), g['loss_L'], rtol=2e-2)
    gnL = np.array([float(pd[k].grad.float().norm()) for k in g['grad_names_L']])
    assert np.allclose(gnL, g['grad_norms_L'], rtol=5e-2), (gnL, g['grad_norms_L'])
    assert pd['bbox_head.cls_convs.0.conv.weight'].grad is None


def test_gradients_vs_oracle_directionally(built):
    """Cosine similarity of full gradient tensors (HIP bf16 vs oracle fp32) for a spread of layers."""
    model, sd0 = built
    sd = {k: v.clone() for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.')):
            v.requires_grad_(True)
    H = W = 128
    img = synth.images(2, H, W)
    gtb, gtl = synth.random_gts(2, H, W, seed=24, gmin=1, gmax=3)
    torch.set_num_threads(8)
    o = omodel.train_step(sd, img, gtb, gtl)
    o['loss'].backward()
    data = dict(img=img.cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    out, *_ = model.train_step(data, Labeled=True, Pseudo=False)
    model.zero_grad()
    out['loss'].backward()
    torch.cuda.synchronize()
    pd = dict(model.named_parameters())
    for k in ['backbone.layer2.0.conv1.weight', 'backbone.layer2.0.bn1.weight', 'backbone.layer2.0.bn1.bias', 'backbone.layer3.5.conv2.weight',
              'backbone.layer4.2.bn3.weight', 'neck.lateral_convs.1.conv.weight', 'neck.fpn_convs.0.conv.bias',
              'bbox_head.cls_convs.2.conv.weight', 'bbox_head.reg_convs.0.conv.bias', 'bbox_head.retina_cls.weight',
              'bbox_head.retina_cls.bias', 'bbox_head.retina_reg.weight']:
        a, b = pd[k].grad.float().cpu().flatten(), sd[k].grad.flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.995, (k, cos)
        assert abs(float(a.norm() / b.norm()) - 1) < 5e-2, (k, float(a.norm()), float(b.norm()))


def test_optimizer_step_reaches_the_packed_weights(built):
    """FusedSGD updates parameters through raw pointers; the derived tensors (packed bf16 weights, folded BN) must follow.
    After one optimizer step the model's loss on the same batch must equal the loss of a FRESH model built from the updated
    state_dict (nothing cached), and differ from the loss before the step."""
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    from aod_meh_hua_amd.optim import FusedSGD
    model, sd = built
    model.load_state_dict(sd, strict=True)
    H = W = 128
    gtb, gtl = synth.random_gts(2, H, W, seed=24, gmin=1, gmax=3)
    data = dict(img=synth.images(2, H, W).cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb],
                gt_labels=[l.cuda() for l in gtl])
    opt = FusedSGD([p for p in model.parameters() if p.requires_grad], lr=0.05, momentum=0.9, weight_decay=1e-4)
    out0, *_ = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad()
    out0['loss'].backward()
    opt.step()
    out1, *_ = model.train_step(data, Labeled=True, Pseudo=False)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    fresh = build_detector(cfg.model)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
    fresh = fresh.cuda().train()
    out2, *_ = fresh.train_step(data, Labeled=True, Pseudo=False)
    l0, l1, l2 = float(out0['loss'].detach()), float(out1['loss'].detach()), float(out2['loss'].detach())
    model.load_state_dict(sd, strict=True)
    assert abs(l1 - l0) > 1e-3 * abs(l0), (l0, l1)
    assert l1 == l2, (l1, l2)


def test_train_step_with_an_image_without_ground_truth(built):
    """Edge case of the reference's assigner (max_iou_assigner.py:145-161, num_gts == 0): every anchor of that image is background,
    num_total_samples counts max(num_pos, 1) per image.  HIP path vs the fp32 oracle."""
    model, sd0 = built
    model.load_state_dict(sd0, strict=True)
    sd = {k: v.clone() for k, v in sd0.items()}
    H = W = 128
    img = synth.images(2, H, W, seed=77)
    gtb, gtl = synth.random_gts(2, H, W, seed=78, gmin=2, gmax=3)
    gtb[1], gtl[1] = torch.zeros(0, 4), torch.zeros(0, dtype=torch.long)
    torch.set_num_threads(8)
    o = omodel.train_step(sd, img, gtb, gtl)
    data = dict(img=img.cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    assert int(head_out[8]) == o['targets']['num_total_pos']
    assert np.allclose(float(out['loss'].detach()), float(o['loss']), rtol=2e-2), (float(out['loss']), float(o['loss']))
    lab = torch.cat([l.reshape(2, -1) for l in head_out[4]], 1).cpu()
    assert bool((lab[1] == 20).all()) and int((lab[0] < 20).sum()) > 0        # image 1: background everywhere
    model.zero_grad()
    out['loss'].backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_grouped_tower_launch_equals_the_separate_towers(built, bf16_mode):
    """Scoring pass: the cls / reg / evidence towers advance together, one grouped launch per depth (aod_conv2d_grouped, 256 x 256 tiles when
    the three towers' tiles fill whole CU rounds); same K order per output element -> bit-identical to the three separate stacks."""
    model, sd = built
    model.load_state_dict(sd, strict=True)
    head = model.bbox_head
    for B, H in ((2, 128), (16, 512)):
        g = synth.gen(77 + B)
        feats = [torch.randn(B, 256, max(H // s, 1), max(H // s, 1), generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
                 for s in (8, 16, 32, 64, 128)]
        with torch.no_grad():
            (c1, r1), l1 = head.forward_all_towers(feats)
            c0, r0 = head.forward(feats)
            l0 = head.forward_L(feats)
        torch.cuda.synchronize()
        for a, b in zip(list(c1) + list(r1) + list(l1), list(c0) + list(r0) + list(l0)):
            assert a.shape == b.shape and torch.equal(a, b)
        assert float(c1[0].float().abs().mean()) > 0


def test_space_to_depth_stem_equals_the_7x7_stem(built, monkeypatch, bf16_mode):
    """The frozen stem as a 4x4 / stride-1 conv over the space-to-depth image against the 7x7 / stride-2 form of the same kernel and
    against torch's fp32 conv of the bf16-rounded operands: same products, regrouped -- only the fp32 summation order differs."""
    import torch.nn.functional as F
    model, sd = built
    model.load_state_dict(sd, strict=True)
    bb = model.backbone
    from aod_meh_hua_amd import functional as AF
    for B, H, W in ((2, 128, 160), (3, 62, 34)):
        img = synth.images(B, H, W).cuda()
        assert AF.stem_s2d_applies(img, bb.conv1, bb.norm1)
        with torch.no_grad():
            y1 = AF.stem_conv_s2d(img, bb.conv1, bb.norm1)
            monkeypatch.setenv('AOD_STEM_S2D', '0')
            assert not AF.stem_s2d_applies(img, bb.conv1, bb.norm1)
            y0 = bb.conv1(AF.image_to_nhwc(img, 8), bn=bb.norm1, relu=True)
            monkeypatch.delenv('AOD_STEM_S2D')
            bn = bb.norm1
            w = bb.conv1.weight.detach().bfloat16().float()
            z = F.conv2d(img.bfloat16().float(), w, None, 2, 3)
            sc = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
            ref = torch.relu(z * sc[None, :, None, None] + (bn.bias.detach() - bn.running_mean * sc)[None, :, None, None])
        torch.cuda.synchronize()
        assert y1.shape == y0.shape == ref.shape
        scale = float(ref.abs().max())
        assert float((y1.float() - ref).abs().max()) <= 1e-2 * scale          # bf16 output rounding (2^-8 relative) + summation order
        assert float((y1.float() - y0.float()).abs().max()) <= 1e-2 * scale
        assert float(y1.float().abs().mean()) > 1e-3
        # ... and with the max-pool fused behind it (aod_stem_pool_fwd: the conv output stays in LDS): identical bits
        with torch.no_grad():
            pooled = AF.stem_pool_s2d(img, bb.conv1, bb.norm1)
            sep = AF.max_pool_3x3_s2(y1)
        torch.cuda.synchronize()
        assert pooled.shape == sep.shape and torch.equal(pooled, sep)
        assert torch.equal(pooled.float(), F.max_pool2d(y1.float(), 3, 2, 1))


@pytest.mark.parametrize('stage,planes,shapes', [('layer2', 128, ((1, 8, 16), (2, 21, 37), (3, 64, 64))),
                                                 ('layer3', 256, ((1, 4, 16), (2, 13, 37), (3, 32, 32)))])
def test_fused_bottleneck128_equals_the_three_launch_block(built, monkeypatch, stage, planes, shapes, bf16_mode):
    """aod_bottleneck128_fwd / aod_bottleneck256_fwd (identity blocks of layer2 / layer3: conv1 on the tile halo, filters streamed through
    LDS rings) against the block as three launches of the implicit-GEMM kernel: one-tile, ragged multi-tile and multi-image inputs -- same
    bf16 rounding points and the same K order per output element -> identical bits; the optional intermediates t1 / t2 (training forward)
    equal the three-launch block's conv1 / conv2 outputs."""
    model, sd = built
    model.load_state_dict(sd, strict=True)
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    layer2 = getattr(model.backbone, stage)
    g = synth.gen(6)
    model.eval()
    for B, H, W in shapes:
        x = (torch.randn(B, 4 * planes, H, W, generator=g).relu() * 0.5).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for blk in list(layer2)[1:]:
                monkeypatch.setenv('AOD_FUSE_BOTTLENECK128', '0')
                assert not AF.bottleneck128_applies(blk, x)
                t1_0 = blk.conv1(x, bn=blk.norm1, relu=True)
                t2_0 = blk.conv2(t1_0, bn=blk.norm2, relu=True)
                y0 = blk(x)
                monkeypatch.setenv('AOD_FUSE_BOTTLENECK128', '1')
                assert AF.bottleneck128_applies(blk, x)
                y1 = blk(x)
                bn = lambda n: (n.weight, n.bias, n.running_mean, n.running_var)
                p1 = AF.PREP.get(blk.conv1.weight, bn(blk.norm1), 4 * planes, blk.norm1.eps)
                p2 = AF.PREP.get(blk.conv2.weight, bn(blk.norm2), planes, blk.norm2.eps)
                p3 = AF.PREP.get(blk.conv3.weight, bn(blk.norm3), planes, blk.norm3.eps)
                y2, t1, t2 = ho.bottleneck128_fwd(AF.as_rows(x), B, H, W, p1.wf, p1.scale, p1.shift, p2.wf, p2.scale, p2.shift, p3.wf, p3.scale,
                                                   p3.shift, keep=True)
                torch.cuda.synchronize()
                assert y1.shape == y0.shape and torch.equal(y1, y0) and torch.equal(AF.as_nchw(y2, B, H, W), y0)
                assert torch.equal(AF.as_nchw(t1, B, H, W), t1_0) and torch.equal(AF.as_nchw(t2, B, H, W), t2_0)
                assert float(y0.float().abs().mean()) > 1e-3
                x = y0
    monkeypatch.delenv('AOD_FUSE_BOTTLENECK128')
    model.train()
    xg = torch.randn(1, 4 * planes, 16, 16).cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert not AF.bottleneck128_applies(layer2[1], xg)          # trainable stage under autograd: per-conv launches


def test_fused_bottleneck_equals_the_three_launch_block(built, monkeypatch, bf16_mode):
    """aod_bottleneck64_fwd (conv1 on the tile halo -> LDS, conv2 gathered from LDS, conv3 + residual from LDS: one launch per frozen layer1
    block) against the block as three / four launches of the implicit-GEMM kernel, on a one-tile image, ragged multi-tile images and
    both input widths: same bf16 rounding points and the same K order per output element -> identical bits."""
    model, sd = built
    model.load_state_dict(sd, strict=True)
    from aod_meh_hua_amd import functional as AF
    layer1 = model.backbone.layer1
    g = synth.gen(5)
    for B, H, W in ((1, 16, 16), (2, 50, 37), (3, 33, 64)):
        x = torch.randn(B, 64, H, W, generator=g).relu().cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for blk in layer1:
                monkeypatch.setenv('AOD_FUSE_BOTTLENECK', '0')
                assert not AF.bottleneck64_applies(blk, x)
                y0 = blk(x)
                monkeypatch.setenv('AOD_FUSE_BOTTLENECK', '1')
                assert AF.bottleneck64_applies(blk, x)
                y1 = blk(x)
                torch.cuda.synchronize()
                assert y1.shape == y0.shape and torch.equal(y1, y0)
                assert float(y0.float().abs().mean()) > 1e-3
                x = y0
    monkeypatch.delenv('AOD_FUSE_BOTTLENECK')
    # the trainable stages keep the per-conv launches (their backward needs the intermediates)
    model.train()
    xg = torch.randn(1, 256, 16, 16).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    assert not AF.bottleneck64_applies(model.backbone.layer2[0], xg)


@pytest.mark.parametrize('stage,planes,shape', [('layer2', 128, (2, 21, 37)), ('layer2', 128, (3, 64, 64)), ('layer3', 256, (2, 13, 37)),
                                               ('layer3', 256, (3, 32, 32))])
def test_fused_bottleneck_backward_equals_the_three_dgrad_launches(built, monkeypatch, stage, planes, shape, bf16_mode):
    """aod_bottleneck_bwd (dgrad chain of an identity block: the three products with the mask / skip-gradient / column-sum epilogues of the
    dgrad launches, intermediates in LDS) against the three launches, through autograd over a whole stage (downsample block + identity
    blocks, so that the ActSlot / skip-gradient hand-overs at both ends of every chain are exercised): input gradient and conv weight
    gradients identical bits; BN gradients (fp32 atomics in both forms) within 1e-5 of their scale."""
    model, sd = built
    model.load_state_dict(sd, strict=True)
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    layer = getattr(model.backbone, stage)
    model.train()                                         # norm_eval=True keeps BN in eval mode (resnet.py:630-640)
    B, H, W = shape
    cin = layer[0].conv1.in_channels
    g = synth.gen(9)
    x0 = (torch.randn(B, cin, 2 * H, 2 * W, generator=g).relu() * 0.5).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(B, 4 * planes, H, W, generator=g) * 0.1).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    params = [q for q in layer.parameters() if q.requires_grad]
    res, calls = {}, {}
    orig = ho.bottleneck_bwd
    for mode in ('0', '1'):
        monkeypatch.setenv('AOD_FUSE_BOTTLENECK_BWD', mode)
        n = [0]
        monkeypatch.setattr(ho, 'bottleneck_bwd', lambda *a, **k: (n.__setitem__(0, n[0] + 1), orig(*a, **k))[1])
        x = x0.clone().requires_grad_(True)
        y = layer(x)
        grads = torch.autograd.grad(y, [x] + params, gy)
        torch.cuda.synchronize()
        res[mode], calls[mode] = [t.clone() for t in grads], n[0]
    assert calls['0'] == 0 and calls['1'] == len(layer) - 1          # every identity block took the fused chain
    names = ['x'] + [n_ for n_, q in layer.named_parameters() if q.requires_grad]
    for n_, a, b in zip(names, res['0'], res['1']):
        if a.dim() == 4:
            assert torch.equal(a, b), n_
        else:
            scale = float(a.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-7, (n_, float((a - b).abs().max()), scale)
    assert float(res['1'][0].float().abs().mean()) > 0


def test_meh_tower_forward_riding_with_the_cls_reg_launches_is_identical(built, monkeypatch):
    """Training: the MEH tower's forward (forward_L on the detached pyramid, Lambda_L2.py:96-103) is computed inside the grouped cls / reg
    tower launches of the main forward (3 x 341 tiles = whole rounds of the CUs) and train_step_L only records its autograd nodes
    (conv_bn_act(pre=...)).  Same launches' arithmetic -> the MEH loss and every MEH gradient are bit-identical to the separate forward; a
    forward_L on a different pyramid ignores the stored outputs."""
    model, sd = built
    from aod_meh_hua_amd import hipops as ho
    H, W = 128, 160
    img = synth.images(2, H, W, seed=41).cuda()
    gtb, gtl = synth.random_gts(2, H, W, seed=42, gmin=1, gmax=3)
    data = dict(img=img, img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    head = model.bbox_head
    Lp = [p for n in head.L_names for p in getattr(head, n).parameters()]
    res, launches = {}, {}
    orig = ho.conv2d_rows_grouped
    for mode in ('0', '1'):
        model.load_state_dict(sd, strict=True)
        model.train()
        monkeypatch.setenv('AOD_MEH_RIDER', mode)
        groups = []
        monkeypatch.setattr(ho, 'conv2d_rows_grouped', lambda xs, *a, **k: (groups.append(len(xs)), orig(xs, *a, **k))[1])
        model.zero_grad(set_to_none=True)
        out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        out['loss'].backward()
        outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
        outL['loss'].backward()
        torch.cuda.synchronize()
        res[mode] = [outL['loss'].detach().clone()] + [p.grad.detach().clone() for p in Lp]
        launches[mode] = groups
    assert launches['0'] == [2] * len(head.cls_convs) and launches['1'] == [3] * len(head.cls_convs)
    for a, b in zip(res['0'], res['1']):
        if a.dim() >= 2:
            assert torch.equal(a, b)                    # weight gradients: deterministic slab sums of identical operands
        else:                                           # loss / bias gradients: fp32 atomics (arrival order) in both forms
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(a.abs().max()) + 1e-12)
    assert float(res['1'][0]) > 0 and all(float(g.abs().max()) > 0 for g in res['1'][1:])
    # stored outputs are tied to the pyramid they were computed on
    monkeypatch.setenv('AOD_MEH_RIDER', '1')
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    other = [f.detach().clone() for f in feat_out]
    a = head.forward_L(other)
    assert head._L_pre is None
    b = head.forward_L(other)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize('shape', [(16, 32, 32), (2, 13, 37), (8, 50, 84), (1, 4, 16), (3, 7, 129)])
def test_register_streamed_bottleneck256_equals_the_ring_form_repeatedly(shape, bf16_mode):
    """aod_bottleneck256f_fwd (fragment-major filter images from aod_frag_pack, filters streamed global -> registers, barrier-free conv2 / conv3
    K loops) against aod_bottleneck256_fwd (row-major packs through LDS rings) on random operands, five launches each: y, t1 and t2 identical
    bits every time (the kernel keeps many loads in flight per wave; a wait that lets one through early shows up as run-to-run garbage)."""
    import ctypes as C
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import call, ptr, stream, lib
    B, H, W = shape
    P, C4 = 256, 1024
    g = torch.Generator(device='cuda').manual_seed(3)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    w = [(rnd(P, C4) * 0.05).bfloat16(), (rnd(P, 9 * P) * 0.03).bfloat16(), (rnd(C4, P) * 0.05).bfloat16()]
    sb = [(torch.rand(n, device='cuda', generator=g) + 0.5, rnd(n) * 0.1) for n in (P, P, C4)]

    class Rec(C.Structure):
        _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int32), ('K', C.c_int32), ('blk0', C.c_int32), ('pad_', C.c_int32)]
    assert lib.aod_frag_pack_item_bytes() == C.sizeof(Rec)
    frags = [torch.empty(t.numel(), dtype=torch.bfloat16, device='cuda') for t in w]
    recs, blk = (Rec * 3)(), 0
    for r, t, f in zip(recs, w, frags):
        r.src, r.dst, r.rows, r.K, r.blk0 = t.data_ptr(), f.data_ptr(), t.shape[0], t.shape[1], blk
        blk += (t.numel() + 2047) // 2048
    tab = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).cuda()
    call('aod_frag_pack', ptr(tab), 3, blk, stream())
    # the image is a permutation of the pack: [step][wave][j][ks][lane][8]
    f0 = frags[0].view(16, 8, 2, 2, 64, 8).cpu()
    lane = torch.arange(64)
    for (s_, w_, j_, ks_) in ((0, 0, 0, 0), (5, 3, 1, 0), (15, 7, 1, 1)):
        rows = 32 * w_ + 8 * ((lane & 15) >> 2) + 4 * j_ + (lane & 3)
        cols = 64 * s_ + 8 * (4 * ks_ + (lane >> 4))
        ref = torch.stack([w[0].cpu()[r, c:c + 8] for r, c in zip(rows.tolist(), cols.tolist())])
        assert torch.equal(f0[s_, w_, j_, ks_], ref)
    for it in range(5):
        x = rnd(B * H * W, C4).relu().bfloat16()
        a = ho.bottleneck128_fwd(x, B, H, W, w[0], *sb[0], w[1], *sb[1], w[2], *sb[2], keep=True)
        b = ho.bottleneck128_fwd(x, B, H, W, frags[0], *sb[0], frags[1], *sb[1], frags[2], *sb[2], keep=True, frag=True)
        torch.cuda.synchronize()
        for name, u, v in zip(('y', 't1', 't2'), a, b):
            assert torch.equal(u, v), (name, it, int((u != v).sum()))
        assert float(a[0].float().abs().mean()) > 1e-3


def test_weight_gradient_queue_survives_an_aborted_backward_pass(built):
    """functional._WgradQueue defers weight-gradient launches to the end of the autograd run.  A run that dies half-way (here: a tensor hook
    raises inside the backbone's backward) never reaches its end-of-pass callback; the next run must neither inherit its queued jobs as its
    own nor go without a callback: its gradients equal those of a run in a fresh state, bit for bit."""
    model, sd = built
    from aod_meh_hua_amd import functional as AF
    H = W = 128
    img = synth.images(2, H, W, seed=77).cuda()
    gtb, gtl = synth.random_gts(2, H, W, seed=78, gmin=1, gmax=3)
    data = dict(img=img, img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    params = [p for p in model.parameters() if p.requires_grad]

    def run(abort):
        model.load_state_dict(sd, strict=True)
        model.train()
        model.zero_grad(set_to_none=True)
        if abort:
            def boom(g):
                raise RuntimeError('boom')
            orig = model.neck.forward

            def hooked(xs):
                xs[-1].register_hook(boom)           # gradient of C5: fires after the head's and the neck's weight gradients were queued
                return orig(xs)
            model.neck.forward = hooked
        try:
            out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        finally:
            if abort:
                del model.neck.forward
        if abort:
            with pytest.raises(RuntimeError, match='boom'):
                out['loss'].backward()
            return None
        out['loss'].backward()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() if p.grad is not None else None for p in params]

    ref = run(False)
    run(True)
    got = run(False)
    assert not AF._WgradQueue.jobs
    for a, b in zip(ref, got):
        assert (a is None) == (b is None)
        if a is not None and a.dim() == 4:
            assert torch.equal(a, b)
        elif a is not None:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(a.abs().max()) + 1e-12)
