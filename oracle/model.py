"""Oracle: RetinaNet-R50/R101-FPN + MEH functional forward on a reference-keyed state_dict,
train_step / train_step_L / scoring restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch fp32 CPU ops
(F.conv2d / F.batch_norm / F.max_pool2d / F.interpolate) -- the same ATen kernels the
reference's nn.Modules dispatch to on CPU -- so it is also the `cpu_baseline` ("port").
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import detect, geometry, hua, losses

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
STRIDES = (8, 16, 32, 64, 128)

# Parity diagnostics (tests/test_gpu_precision_x3.py, tools/dbg/x3_grad_table.py): evaluate the network with the ReLU SIGN PATTERN of another
# run.  A ReLU whose input lies within the other implementation's rounding error of zero may open on one side and stay shut on the other; the
# values differ by that rounding error, but the GRADIENTS differ by the whole upstream gradient of that element.  With `relu_masks` the oracle's
# `relu(z)` at a keyed site becomes `z * mask` (mask: bool tensor of z's shape), so the two backward passes walk the same piecewise-linear
# branch.  Keys: 'backbone.bn1'; '<block>.bn1' / '.bn2' / '.out'; 'bbox_head.<tower>.<i>@<level>'; 'bbox_head.retina_L@<level>'.
_MASKS = None


class relu_masks:
    def __init__(self, masks):
        self.masks = masks

    def __enter__(self):
        global _MASKS
        self.prev, _MASKS = _MASKS, self.masks

    def __exit__(self, *exc):
        global _MASKS
        _MASKS = self.prev


def _relu(z, key):
    m = _MASKS.get(key) if _MASKS is not None else None
    if m is None:
        return F.relu(z)
    assert m.shape == z.shape, (key, m.shape, z.shape)
    return z * m.to(z.dtype)


def state_dict_spec(depth=50, num_classes=20, num_anchors=9):
    """Ordered (key, shape) list of the reference model's state_dict (SURVEY 8b 'Checkpoint keys';
    observed on the reference model, asserted equal in tests/test_oracle_golden.py)."""
    spec = []

    def bn(prefix, c):
        spec.extend([(prefix + '.weight', (c,)), (prefix + '.bias', (c,)), (prefix + '.running_mean', (c,)),
                     (prefix + '.running_var', (c,)), (prefix + '.num_batches_tracked', ())])
    spec.append(('backbone.conv1.weight', (64, 3, 7, 7)))
    bn('backbone.bn1', 64)
    inpl = 64
    for li, nb in enumerate(ARCH[depth]):
        planes = 64 * 2 ** li
        for bi in range(nb):
            p = f'backbone.layer{li + 1}.{bi}'
            spec.append((p + '.conv1.weight', (planes, inpl, 1, 1)))
            bn(p + '.bn1', planes)
            spec.append((p + '.conv2.weight', (planes, planes, 3, 3)))
            bn(p + '.bn2', planes)
            spec.append((p + '.conv3.weight', (planes * 4, planes, 1, 1)))
            bn(p + '.bn3', planes * 4)
            if bi == 0:
                spec.append((p + '.downsample.0.weight', (planes * 4, inpl, 1, 1)))
                bn(p + '.downsample.1', planes * 4)
            inpl = planes * 4
    for i, c in enumerate((512, 1024, 2048)):
        spec.append((f'neck.lateral_convs.{i}.conv.weight', (256, c, 1, 1)))
        spec.append((f'neck.lateral_convs.{i}.conv.bias', (256,)))
    for i in range(5):
        cin = 2048 if i == 3 else 256
        spec.append((f'neck.fpn_convs.{i}.conv.weight', (256, cin, 3, 3)))
        spec.append((f'neck.fpn_convs.{i}.conv.bias', (256,)))
    for t in ('cls_convs', 'reg_convs', 'L_convs'):
        for i in range(4):
            spec.append((f'bbox_head.{t}.{i}.conv.weight', (256, 256, 3, 3)))
            spec.append((f'bbox_head.{t}.{i}.conv.bias', (256,)))
    for n, c in (('retina_cls', num_anchors * num_classes), ('retina_reg', num_anchors * 4), ('retina_L', num_anchors)):
        spec.append((f'bbox_head.{n}.weight', (c, 256, 3, 3)))
        spec.append((f'bbox_head.{n}.bias', (c,)))
    return spec


def seeded_state_dict(depth=50, num_classes=20, cls_bias=0.0, bn3_gamma=0.3):
    """The fixed weight recipe shared by golden generation and the build (the real
    R50 checkpoint is 156 MB, so both sides REGENERATE it): tensor idx in state_dict order ->
    torch.Generator().manual_seed(20+idx); conv weights Kaiming-normal (std sqrt(2/fan_in)) except
    the three `retina_*` prediction convs N(0, 0.01) (the reference's head init std,
    Lambda_L2.py:28-29); conv biases 0 (+cls_bias on retina_cls); BN beta=0, mean=0, var=1,
    gamma=1 except bn3 -> bn3_gamma (0.3; the 23-block layer3 of R101 needs ~0.12) and downsample.1 -> 0.7 so the residual sum
    keeps O(1) variance."""
    sd = OrderedDict()
    for idx, (k, shp) in enumerate(state_dict_spec(depth, num_classes)):
        g = torch.Generator().manual_seed(20 + idx)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif len(shp) == 4:
            s = 0.01 if 'retina_' in k else float(np.sqrt(2.0 / (shp[1] * shp[2] * shp[3])))
            sd[k] = torch.randn(shp, generator=g) * s
        elif k.endswith('running_var'):
            sd[k] = torch.ones(shp)
        elif k.endswith('.weight'):            # BN gamma
            sd[k] = torch.full(shp, bn3_gamma if '.bn3.' in k else (0.7 if 'downsample.1' in k else 1.0))
        else:
            sd[k] = torch.zeros(shp)
    sd['bbox_head.retina_cls.bias'] += cls_bias
    return sd


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, 1e-5)


def backbone(sd, img, depth=50):
    """ResNet.forward / Bottleneck.forward (backbones/resnet.py:630-645, :262-301), pytorch style
    (stride on conv2), BN in eval mode (norm_eval=True, :647-656)."""
    x = F.conv2d(img, sd['backbone.conv1.weight'], None, 2, 3)
    x = _relu(_bn(x, sd, 'backbone.bn1'), 'backbone.bn1')
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nb in enumerate(ARCH[depth]):
        for bi in range(nb):
            p = f'backbone.layer{li + 1}.{bi}'
            stride = 2 if (bi == 0 and li > 0) else 1
            o = _relu(_bn(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'), p + '.bn1')
            o = _relu(_bn(F.conv2d(o, sd[p + '.conv2.weight'], None, stride, 1), sd, p + '.bn2'), p + '.bn2')
            o = _bn(F.conv2d(o, sd[p + '.conv3.weight']), sd, p + '.bn3')
            idt = x
            if bi == 0:
                idt = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride), sd, p + '.downsample.1')
            x = _relu(o + idt, p + '.out')
        outs.append(x)
    return outs


def fpn(sd, feats):
    """FPN.forward (necks/fpn.py:151-202), start_level=1, add_extra_convs='on_input', num_outs=5."""
    lat = [F.conv2d(feats[i + 1], sd[f'neck.lateral_convs.{i}.conv.weight'], sd[f'neck.lateral_convs.{i}.conv.bias'])
           for i in range(3)]
    for i in (2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(lat[i], sd[f'neck.fpn_convs.{i}.conv.weight'], sd[f'neck.fpn_convs.{i}.conv.bias'], 1, 1)
            for i in range(3)]
    outs.append(F.conv2d(feats[3], sd['neck.fpn_convs.3.conv.weight'], sd['neck.fpn_convs.3.conv.bias'], 2, 1))
    outs.append(F.conv2d(outs[-1], sd['neck.fpn_convs.4.conv.weight'], sd['neck.fpn_convs.4.conv.bias'], 2, 1))
    return outs


def _tower(sd, x, name, pred, relu_out=False, lvl=0):
    for i in range(4):
        x = _relu(F.conv2d(x, sd[f'bbox_head.{name}.{i}.conv.weight'], sd[f'bbox_head.{name}.{i}.conv.bias'], 1, 1), f'bbox_head.{name}.{i}@{lvl}')
    y = F.conv2d(x, sd[f'bbox_head.{pred}.weight'], sd[f'bbox_head.{pred}.bias'], 1, 1)
    return _relu(y, f'bbox_head.{pred}@{lvl}') if relu_out else y


def head_forward(sd, feats):
    """Lambda_L2Net.forward_single (Lambda_L2.py:85-94)."""
    return ([_tower(sd, f, 'cls_convs', 'retina_cls', lvl=l) for l, f in enumerate(feats)],
            [_tower(sd, f, 'reg_convs', 'retina_reg', lvl=l) for l, f in enumerate(feats)])


def head_forward_L(sd, feats):
    """Lambda_L2Net.forward_single_L (Lambda_L2.py:96-103)."""
    return [_tower(sd, f, 'L_convs', 'retina_L', relu_out=True, lvl=l) for l, f in enumerate(feats)]


def nhwc_flat(x, c):
    """[B, A*c, h, w] -> [B, h*w*A, c]  (the permute(0,2,3,1).reshape of Lambda_L2.py:114)."""
    B = x.shape[0]
    return x.permute(0, 2, 3, 1).reshape(B, -1, c)


def anchors_for(featmap_sizes, pad_shapes):
    base = geometry.gen_base_anchors(STRIDES)
    mlvl = geometry.grid_anchors(base, featmap_sizes, STRIDES)
    flags = [geometry.valid_flags(featmap_sizes, STRIDES, ps, [9] * 5) for ps in pad_shapes]
    return mlvl, flags


def train_step(sd, img, gt_bboxes, gt_labels, pad_shapes=None, depth=50, num_classes=20):
    """SSLBase_L_Detector.train_step -> forward_train -> L_AnchorHead.loss -> _parse_losses
    (SSL_Lambda.py:156-162; SSL_L_single_stage.py:51-62; L_anchor_head.py:290-320).
    Returns dict(loss, loss_cls[5], loss_bbox[5], loss_noR[5] (per-anchor), feats, targets)."""
    B, _, H, W = img.shape
    pad_shapes = pad_shapes or [(H, W, 3)] * B
    feats = fpn(sd, backbone(sd, img, depth))
    cls, reg = head_forward(sd, feats)
    sizes = [tuple(f.shape[-2:]) for f in feats]
    mlvl, flags = anchors_for(sizes, pad_shapes)
    tg = geometry.get_targets(mlvl, flags, gt_bboxes, gt_labels, num_classes)
    n = tg['num_total_pos']
    lc, lb, lnr = [], [], []
    for l in range(5):
        a, b, c = losses.loss_single(nhwc_flat(cls[l], num_classes).reshape(-1, num_classes),
                                     nhwc_flat(reg[l], 4).reshape(-1, 4), tg['labels'][l], tg['label_weights'][l],
                                     tg['bbox_targets'][l], tg['bbox_weights'][l], n)
        lc.append(a), lb.append(b), lnr.append(c)
    loss, _ = losses.parse_losses(dict(loss_cls=lc, loss_bbox=lb, loss_noR=lnr))
    return dict(loss=loss, loss_cls=lc, loss_bbox=lb, loss_noR=lnr, feats=feats, targets=tg, cls=cls, reg=reg)


def train_step_L(sd, feats, loss_noR, targets):
    """train_step_L -> forward_train_L -> loss_L -> loss_single_L (SSL_Lambda.py:164-168;
    Lambda_L2.py:62-64, 235-241; L_anchor_head.py:322-327) on DETACHED feats / losses."""
    Ls = head_forward_L(sd, [f.detach() for f in feats])
    ll = [losses.meh_loss_single(nhwc_flat(Ls[l], 1).reshape(-1), loss_noR[l].detach(), targets['bbox_weights'][l])
          for l in range(5)]
    return dict(loss=sum(ll), loss_L=ll)


def sgd_step(params, grads, bufs, lr=1e-3, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD semantics (apis/train_Lambda.py:54,59-61): d = g + wd*p; buf = m*buf + d
    (first step buf = d); p -= lr*buf."""
    for k in params:
        if grads.get(k) is None:
            continue
        d = grads[k] + weight_decay * params[k]
        if bufs.get(k) is None:
            bufs[k] = d.clone()
        else:
            bufs[k].mul_(momentum).add_(d)
        params[k].sub_(lr * bufs[k])


def score_images(sd, img, img_shapes=None, scale_factors=None, sampler='torch', seed=20, image_ids=None,
                 depth=50, num_classes=20, uPool2='objectSum_scaleMax_classSum', heads=None):
    """simple_test(isEval=False, uPool='Entropy_NMS') (SSL_L_single_stage.py:68-98; Lambda_L2.py:398-420,
    254-384): forward + MEH forward + pre-NMS + NMS + GetObjectIdx + ComputeObjUnc + aggregate.
    `heads` = (cls, reg, L) lists may be injected (planted-logit scoring)."""
    B, _, H, W = img.shape
    img_shapes = img_shapes or [(H, W, 3)] * B
    scale_factors = scale_factors or [np.ones(4, np.float32)] * B
    if heads is None:
        with torch.no_grad():
            feats = fpn(sd, backbone(sd, img, depth))
            cls, reg = head_forward(sd, feats)
            Ls = head_forward_L(sd, feats)
    else:
        cls, reg, Ls = heads
    sizes = [tuple(c.shape[-2:]) for c in cls]
    mlvl = geometry.grid_anchors(geometry.gen_base_anchors(STRIDES), sizes, STRIDES)
    pre = detect.pre_nms([nhwc_flat(c, num_classes) for c in cls], [nhwc_flat(r, 4) for r in reg],
                         [nhwc_flat(l, 1)[..., 0] for l in Ls], mlvl, img_shapes, scale_factors)
    dets, pos = [], []
    for b in range(B):
        d, lab, keep, inds = detect.multiclass_nms(pre['cat_boxes'][b], pre['cat_scores'][b])
        dets.append((d, lab, keep))
        pos.append(detect.get_object_idx(d, pre['cat_boxes'][b]))
    level_offsets = np.concatenate([[0], np.cumsum([a.shape[0] for a in mlvl])[:-1]])
    bins, pairs = hua.compute_obj_unc(pre, pos, sampler=sampler, seed=seed, image_ids=image_ids,
                                      level_offsets=level_offsets)
    unc = hua.aggregate_obj_scale_unc(bins, uPool2)
    return dict(unc=unc, dets=dets, pos=pos, pre=pre, bins=bins, pairs=pairs)
