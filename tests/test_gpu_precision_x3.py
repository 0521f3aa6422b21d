"""GPU: the reference-precision mode `bf16x3` (aod_meh_hua_amd/functional.py set_precision, csrc/conv.hip "X3": head/tail-split activations, gradients and filters in the
X-layout, three MFMAs per product inside the SAME implicit-GEMM / dgrad / wgrad kernels with their fused epilogues) against the golden values
the REFERENCE produced (tests/golden/train_step.npz) and the fp32 oracle's gradients: fp32-level agreement, where the bf16 mode's residuals
(operand rounding, not logic) are two to three orders of magnitude larger."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    sd = omodel.seeded_state_dict()
    model.load_state_dict(sd, strict=True)
    return model.cuda().train(), sd


def _run(model, prec):
    from aod_meh_hua_amd import functional as AF
    g = np.load(os.path.join(G, 'train_step.npz'))
    H = W = 128
    gtb, gtl = synth.random_gts(2, H, W, seed=24, gmin=1, gmax=3)
    data = dict(img=synth.images(2, H, W).cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    AF.set_precision(prec)
    try:
        out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        model.zero_grad()
        out['loss'].backward()
        pd = dict(model.named_parameters())
        grads = {k: pd[k].grad.detach().float().cpu().clone() for k in pd if pd[k].grad is not None}
        lossL = model.train_step_L(prev, head_out, feat_out)
        model.zero_grad()
        lossL['loss'].backward()
        gradsL = {k: pd[k].grad.detach().float().cpu().clone() for k in pd if pd[k].grad is not None}
        feat4 = AF.x3_to_f32(feat_out[4], 256).cpu().numpy()          # (an X-layout tensor in the bf16x3 mode)
        torch.cuda.synchronize()
    finally:
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))
    dev = dict(
        loss=abs(float(out['loss']) - float(g['loss'])) / abs(float(g['loss'])),
        log_vars=rel([float(out['log_vars'][k]) for k in ('loss_cls', 'loss_bbox', 'loss_noR')], g['log_vars']),
        feat_l4=rel(feat4, g['feat_l4']),
        cls_l3=rel(head_out[1][3].detach().float().cpu().numpy(), g['cls_l3']),
        loss_noR_l4=rel(prev[4].cpu().numpy(), g['loss_noR_l4']),
        loss_L=abs(float(lossL['loss']) - float(g['loss_L'])) / abs(float(g['loss_L'])),
        grad_norms=float(np.abs(np.array([float(grads[k].norm()) for k in g['grad_names']]) / g['grad_norms'] - 1).max()),
        grad_norms_L=float(np.abs(np.array([float(gradsL[k].norm()) for k in g['grad_names_L']]) / g['grad_norms_L'] - 1).max()))
    return dev, grads, int(head_out[8])


def test_bf16x3_mode_collapses_the_deviation_from_the_reference_golden():
    model, sd0 = _model()
    g = np.load(os.path.join(G, 'train_step.npz'))
    dev16, grads16, n16 = _run(model, 'bf16')
    dev3, grads3, n3 = _run(model, 'bf16x3')
    print('deviation from the reference golden  bf16:', {k: f'{v:.2e}' for k, v in dev16.items()})
    print('deviation from the reference golden  bf16x3:', {k: f'{v:.2e}' for k, v in dev3.items()})
    assert n16 == n3 == int(g['num_total_samples'])
    # same kernels, split operands: fp32-level agreement with the reference (the reference's own CPU summation order differs at ~1e-6)
    assert dev3['loss'] < 1e-4 and dev3['log_vars'] < 1e-4 and dev3['loss_L'] < 2e-4, dev3
    assert dev3['feat_l4'] < 2e-4 and dev3['cls_l3'] < 2e-4 and dev3['loss_noR_l4'] < 5e-4, dev3
    # (gradient norms: bounded by the ReLU sign flips of elements within the forward rounding error of zero -- next test; the arithmetic
    # of the backward pass itself holds 5e-5 once both sides walk the same branch)
    assert dev3['grad_norms'] < 2e-3 and dev3['grad_norms_L'] < 2e-3, dev3
    # and the product mode's residuals were rounding: they shrink by more than an order of magnitude in every quantity that is
    # measurably off in bf16
    for k, v in dev16.items():
        if v > 2e-3:
            assert dev3[k] < v / 10, (k, v, dev3[k])
    # gradient DIRECTIONS against the fp32 oracle
    sd = {k: v.clone() for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.')):
            v.requires_grad_(True)
    gtb, gtl = synth.random_gts(2, 128, 128, seed=24, gmin=1, gmax=3)
    torch.set_num_threads(8)
    o = omodel.train_step(sd, synth.images(2, 128, 128), gtb, gtl)
    o['loss'].backward()
    worst16 = worst3 = 0.0
    for k in ['backbone.layer2.0.conv1.weight', 'backbone.layer2.0.bn1.weight', 'backbone.layer3.5.conv2.weight', 'backbone.layer4.2.bn3.weight',
              'neck.lateral_convs.1.conv.weight', 'neck.fpn_convs.4.conv.weight', 'bbox_head.cls_convs.2.conv.weight', 'bbox_head.retina_cls.weight',
              'bbox_head.retina_cls.bias', 'bbox_head.retina_reg.weight']:
        b = sd[k].grad.flatten()
        e16 = float((grads16[k].flatten() - b).norm() / b.norm())
        e3 = float((grads3[k].flatten() - b).norm() / b.norm())
        worst16, worst3 = max(worst16, e16), max(worst3, e3)
        assert e3 < 5e-3 and e3 < e16 / 5, (k, e3, e16)          # (the dropped tail x tail term and fp32 summation order remain)
    print('relative gradient error vs the fp32 oracle: bf16 worst', worst16, ' bf16x3 worst', worst3)
    assert worst3 < worst16 / 8


def test_bf16x3_scoring_pass_matches_the_fp32_oracle_and_replays_as_a_graph():
    """the reference-precision mode covers the SCORING phase too (forward, MEH forward, top-k, NMS, HUA) and both phases
    replay as HIP graphs.  Full model on seeded weights with a trained-like (scaled) classification head: in bf16x3 the detections agree with
    the fp32 CPU oracle to 1e-4, in the same order, and the image scores follow (Philox sampler on both sides); the bf16 product mode is printed beside it."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.graphs import GraphedScore
    model, sd0 = _model()
    H = W = 128
    img = synth.images(2, H, W, seed=31)
    torch.set_num_threads(8)
    sd = {k: v.clone() for k, v in sd0.items()}
    with torch.no_grad():       # calibrate on the ORACLE's logits: scale retina_cls until ~1 % of the anchors are foreground (> 0.3)
        cls, _ = omodel.head_forward(sd, omodel.fpn(sd, omodel.backbone(sd, img)))
        x = torch.cat([omodel.nhwc_flat(c, 20) for c in cls], 1)
        lo, hi = 1.0, 1e5
        for _ in range(40):
            mid = (lo * hi) ** 0.5
            lo, hi = (mid, hi) if float((torch.softmax(x * mid, -1).amax(-1) > 0.3).float().mean()) < 0.01 else (lo, mid)
        k = float(np.float32(hi))
        for key in ('bbox_head.retina_cls.weight', 'bbox_head.retina_cls.bias'):
            sd[key] = sd[key] * k
    model.load_state_dict(sd, strict=True)
    model.eval()
    mt = synth.metas(2, H, W)
    o = omodel.score_images(sd, img, [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt], sampler='philox', seed=20)
    ref_unc = np.array(o['unc'])
    assert (ref_unc > 0).all(), ref_unc
    kw = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
              showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
    ids = torch.arange(2, device='cuda')
    res = {}
    for prec in ('bf16', 'bf16x3'):
        AF.set_precision(prec)
        try:
            with torch.no_grad():
                dets, unc = model(img=[img.cuda()], img_metas=[mt], return_loss=False, image_ids=ids, **kw)
                gs = GraphedScore(model, **kw)
                _, unc_g1 = gs(img.cuda(), mt, ids)
                _, unc_g2 = gs(img.cuda(), mt, ids)          # replay
            torch.cuda.synchronize()
        finally:
            AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
        u = torch.as_tensor(unc).float().cpu().numpy()
        assert np.array_equal(u, torch.as_tensor(unc_g1).float().cpu().numpy()) and np.array_equal(u, torch.as_tensor(unc_g2).float().cpu().numpy())
        res[prec] = (u, dets)
        print(prec, 'unc', u, 'oracle', ref_unc, 'rel dev', np.abs(u - ref_unc) / ref_unc)
    u3, d3 = res['bf16x3']
    for b in range(2):
        od, olab, _ = o['dets'][b]
        gd, glab = d3[b]
        gd, glab = torch.as_tensor(gd).float().cpu(), torch.as_tensor(glab).cpu()
        assert gd.shape[0] == od.shape[0]
        # the same detections: every oracle detection has a HIP detection of its class with the same box (1e-3 px) and score (1e-4);
        # the ORDER of detections whose scores differ by < 1e-5 may swap, and the last few of the 100 kept may trade places with the first
        # few not kept
        hit = 0
        for k in range(od.shape[0]):
            same = (glab.long() == int(olab[k])) & ((gd[:, :4] - od[k, :4]).abs().amax(1) < 1e-2) & ((gd[:, 4] - od[k, 4]).abs() < 1e-4)
            hit += int(same.any())
        assert hit >= od.shape[0] - 3, (b, hit, od.shape[0])
    # image scores: a (candidate, object) pair whose IoU sits within 1e-5 of the 0.5 gate (Lambda_L2.py:349) may still fall on the other side
    # -- one pair of ~100 moves a score by ~0.3 % --, everything else agrees to 1e-5
    assert np.allclose(u3, ref_unc, rtol=1e-2), (u3, ref_unc)
    assert np.abs(u3 - ref_unc).min() / ref_unc.max() < 1e-4


@pytest.mark.parametrize('size', [128, 256])
def test_bf16x3_gradient_residual_is_relu_sign_flips(size):
    """VERDICT r4 item 2.  The gradients of the reference-precision mode sit 1e-3 (worst tensor 3e-3 ... 5e-3) from the fp32 / fp64 oracle's
    where its losses and features sit 1e-6 ... 1e-5.  Root cause, asserted here over ALL ~170 trainable tensors of both optimizer steps: 16-bit
    operands put ~4 of every million ReLU inputs on the other side of zero than exact arithmetic does (25 of 5.5 M elements at 2 x 128^2); the
    VALUES move by the rounding error, but a flipped element hands its WHOLE upstream gradient on (or blocks it), so a tensor's relative gradient
    error is sqrt(share of the squared gradient carried by flipped elements) -- a property of any arithmetic with 1e-5 forward error in front of
    a ReLU, not of the backward kernels.  With the HIP run's own ReLU sign pattern injected into the fp64 oracle (oracle.model.relu_masks: both
    backward passes then walk the same piecewise-linear branch) every gradient agrees to 1e-4 (measured: worst 3.7e-5 / 4.6e-5, median 3e-5).
    The fp32 oracle against the fp64 one shows the same mechanism at its own, 100 x smaller, flip rate (profiles/r05_x3_grad_error_*.json)."""
    from aod_meh_hua_amd import functional as AF
    from tests.maskcap import capture_relu_masks
    model, sd0 = _model()
    B, H = 2, size
    gtb, gtl = synth.random_gts(B, H, H, seed=24, gmin=1, gmax=3)
    img = synth.images(B, H, H)
    data = dict(img=img.cuda(), img_metas=synth.metas(B, H, H), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    AF.set_precision('bf16x3')
    with capture_relu_masks(model) as masks:
        out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    model.zero_grad()
    out['loss'].backward()
    pd = dict(model.named_parameters())
    grads = {k: pd[k].grad.detach().double().cpu().clone() for k in pd if pd[k].grad is not None}
    with capture_relu_masks(model, masks):
        lossL = model.train_step_L(prev, head_out, feat_out)
    model.zero_grad()
    lossL['loss'].backward()
    gradsL = {k: pd[k].grad.detach().double().cpu().clone() for k in pd if pd[k].grad is not None}
    torch.cuda.synchronize()
    assert len(masks) == 104, len(masks)          # 13 trainable blocks x 3 + (4 x 3 towers + retina_L) x 5 levels

    def oracle(masked):
        sd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.')):
                v.requires_grad_(True)
        import contextlib
        with (omodel.relu_masks(masks) if masked else contextlib.nullcontext()):
            o = omodel.train_step(sd, img.double(), gtb, gtl)
            o['loss'].backward()
            g1 = {k: v.grad.clone() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
            for v in sd.values():
                if v.is_floating_point():
                    v.grad = None
            oL = omodel.train_step_L(sd, o['feats'], o['loss_noR'], o['targets'])
            oL['loss'].backward()
            gL = {k: v.grad.clone() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
        return float(o['loss']), g1, gL

    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    lp, p1, pL = oracle(False)
    lm, m1, mL = oracle(True)
    assert abs(float(out['loss']) - lp) <= 1e-5 * abs(lp) and abs(lm - lp) <= 1e-6 * abs(lp)       # values: the flips move nothing measurable
    raw, msk = [], []
    for gh, gp, gm in ((grads, p1, m1), (gradsL, pL, mL)):
        for k, ref in gm.items():
            if k in gh and float(ref.norm()) > 0:
                raw.append(float((gh[k] - gp[k]).norm() / gp[k].norm()))
                msk.append(float((gh[k] - ref).norm() / ref.norm()))
    assert len(msk) >= 170, len(msk)
    print(f'{B}x{H}^2: {len(msk)} tensors; gradient error vs the fp64 oracle: worst {max(raw):.2e} median {np.median(raw):.2e}; '
          f'with the HIP ReLU sign pattern: worst {max(msk):.2e} median {np.median(msk):.2e}')
    assert max(msk) < 1e-4, max(msk)
    assert max(raw) < 1e-2 and np.median(raw) < 2e-3, (max(raw), np.median(raw))
