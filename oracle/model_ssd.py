"""Oracle: SSD300-VGG16 (and SSD512, `v=V512`: configs/ssd/ssd512_voc.py) + MEH/HUA (BASELINE config 0) as plain functional torch fp32 on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
  mmdet/models/backbones/ssd_vgg.py:12-118 (+ mmcv.cnn.VGG layer order), mmdet/models/necks/ssd_neck.py (extra layers + L2Norm),
  mmdet/core/anchor/anchor_generator.py:460-564 (SSDAnchorGenerator),
  mmdet/models/dense_heads/My_L_ssd_head.py:102-137 (layers), :169-180 (forward / forward_L), :182-215 (loss_single),
  :217-224 + :302-313 (MEH loss), :226-300 (loss), :316-433 (_get_bboxes), :435-482 (ComputeObjUnc).
Pinned by tests/golden/ssd_*.npz and ssd512_*.npz (tools/golden/make_golden_ssd.py runs the reference itself, SSD300 and SSD512)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import detect, geometry, hua, losses

VGG16 = (2, 2, 3, 3, 3)


class Variant:
    """one SSD input size: Config_SSD.py:23-62 (300) or that config with the overrides of configs/ssd/ssd512_voc.py (512)"""

    def __init__(self, input_size, in_ch, num_anchors, strides, ratios, extra, basesize_ratio_range, sizes):
        self.input_size, self.IN_CH, self.NUM_ANCHORS, self.STRIDES, self.RATIOS, self.EXTRA = input_size, in_ch, num_anchors, strides, ratios, extra
        self.basesize_ratio_range, self.SIZES = basesize_ratio_range, sizes


# EXTRA rows: (in, mid, out, kernel, stride, pad) of one extra level (ssd_neck.py:64-88: 1x1 reduce, then kernel x kernel)
V300 = Variant(300, (512, 1024, 512, 256, 256, 256), (4, 6, 6, 6, 4, 4), (8, 16, 32, 64, 100, 300), ([2], [2, 3], [2, 3], [2, 3], [2], [2]),
               ((1024, 256, 512, 3, 2, 1), (512, 128, 256, 3, 2, 1), (256, 128, 256, 3, 1, 0), (256, 128, 256, 3, 1, 0)), (0.15, 0.9),
               (38, 19, 10, 5, 3, 1))
V512 = Variant(512, (512, 1024, 512, 256, 256, 256, 256), (4, 6, 6, 6, 6, 4, 4), (8, 16, 32, 64, 128, 256, 512),
               ([2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]),
               ((1024, 256, 512, 3, 2, 1), (512, 128, 256, 3, 2, 1), (256, 128, 256, 3, 2, 1), (256, 128, 256, 3, 2, 1), (256, 128, 256, 4, 1, 1)),
               (0.1, 0.9), (64, 32, 16, 8, 4, 2, 1))
IN_CH, NUM_ANCHORS, STRIDES, RATIOS = V300.IN_CH, V300.NUM_ANCHORS, V300.STRIDES, V300.RATIOS       # (SSD300 names kept for the existing tests)
CODER_STDS = (0.1, 0.1, 0.2, 0.2)
ASSIGNER = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.0, gt_max_assign_all=False)


def vgg_layers():
    """[(kind, feature_index, args)] in nn.Sequential order: mmcv VGG16 stages with ceil-mode pools, last pool dropped, then
    ssd_vgg.py:60-68 pool5(3,1,1) / fc6 (3x3, dilation 6) / fc7 (1x1)."""
    out, idx, inpl = [], 0, 3
    for i, nb in enumerate(VGG16):
        planes = 64 * 2 ** i if i < 4 else 512
        for _ in range(nb):
            out.append(('conv', idx, (inpl, planes, 3, 1, 1)))
            out.append(('relu', idx + 1, None))
            idx += 2
            inpl = planes
        out.append(('pool', idx, (2, 2, 0, True)))
        idx += 1
    out.pop(-1)
    idx -= 1
    out.append(('pool', idx, (3, 1, 1, False)))
    out.append(('conv', idx + 1, (512, 1024, 3, 6, 6)))
    out.append(('relu', idx + 2, None))
    out.append(('conv', idx + 3, (1024, 1024, 1, 0, 1)))
    out.append(('relu', idx + 4, None))
    return out


def state_dict_spec(num_classes=20, v=V300):
    spec = []
    for kind, idx, a in vgg_layers():
        if kind == 'conv':
            spec += [(f'backbone.features.{idx}.weight', (a[1], a[0], a[2], a[2])), (f'backbone.features.{idx}.bias', (a[1],))]
    spec.append(('neck.l2_norm.weight', (512,)))
    for i, (cin, mid, cout, k, s, p) in enumerate(v.EXTRA):
        spec += [(f'neck.extra_layers.{i}.0.conv.weight', (mid, cin, 1, 1)), (f'neck.extra_layers.{i}.0.conv.bias', (mid,)),
                 (f'neck.extra_layers.{i}.1.conv.weight', (cout, mid, k, k)), (f'neck.extra_layers.{i}.1.conv.bias', (cout,))]
    for name, per in (('cls_convs', num_classes + 1), ('reg_convs', 4), ('L_convs', 1)):
        for l, (c, na) in enumerate(zip(v.IN_CH, v.NUM_ANCHORS)):
            spec += [(f'bbox_head.{name}.{l}.0.weight', (na * per, c, 3, 3)), (f'bbox_head.{name}.{l}.0.bias', (na * per,))]
    return spec


def seeded_state_dict(num_classes=20, v=V300):
    """Weight recipe shared by the golden generator and the build (the VGG16-caffe checkpoint is not available offline):
    tensor idx in state_dict order -> Generator(120+idx); backbone / neck conv weights Kaiming-normal, head conv weights N(0, 0.01),
    biases 0.02*N(0,1) (so bias gradients are exercised), L2Norm weight 20 (ssd_neck.py init)."""
    sd = OrderedDict()
    for idx, (k, shp) in enumerate(state_dict_spec(num_classes, v)):
        g = torch.Generator().manual_seed(120 + idx)
        if k == 'neck.l2_norm.weight':
            sd[k] = torch.full(shp, 20.0)
        elif len(shp) == 4:
            s = 0.01 if k.startswith('bbox_head') else float(np.sqrt(2.0 / (shp[1] * shp[2] * shp[3])))
            sd[k] = torch.randn(shp, generator=g) * s
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.02
    return sd


def backbone(sd, img):
    """ssd_vgg.py:107-118: outputs after features[22] (conv4_3 ReLU) and features[34] (fc7 ReLU)."""
    x, outs = img, []
    for kind, idx, a in vgg_layers():
        if kind == 'conv':
            x = F.conv2d(x, sd[f'backbone.features.{idx}.weight'], sd[f'backbone.features.{idx}.bias'], 1, a[3], a[4])
        elif kind == 'relu':
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, a[0], a[1], a[2], ceil_mode=a[3])
        if idx in (22, 34):
            outs.append(x)
    return outs


def l2norm(x, weight, eps=1e-10):
    """ssd_neck.py L2Norm.forward: x / (sqrt(sum_c x^2) + eps) * weight."""
    xf = x.float()
    norm = xf.pow(2).sum(1, keepdim=True).sqrt() + eps
    return (weight[None, :, None, None].float().expand_as(xf) * xf / norm).type_as(x)


def neck(sd, feats, v=V300):
    outs = [l2norm(feats[0], sd['neck.l2_norm.weight']), feats[1]]
    x = feats[1]
    for i, (cin, mid, cout, k, s, p) in enumerate(v.EXTRA):
        x = F.relu(F.conv2d(x, sd[f'neck.extra_layers.{i}.0.conv.weight'], sd[f'neck.extra_layers.{i}.0.conv.bias']))
        x = F.relu(F.conv2d(x, sd[f'neck.extra_layers.{i}.1.conv.weight'], sd[f'neck.extra_layers.{i}.1.conv.bias'], s, p))      # (k x k from the weight)
        outs.append(x)
    return outs


def head_forward(sd, feats):
    cls = [F.conv2d(f, sd[f'bbox_head.cls_convs.{l}.0.weight'], sd[f'bbox_head.cls_convs.{l}.0.bias'], 1, 1) for l, f in enumerate(feats)]
    reg = [F.conv2d(f, sd[f'bbox_head.reg_convs.{l}.0.weight'], sd[f'bbox_head.reg_convs.{l}.0.bias'], 1, 1) for l, f in enumerate(feats)]
    return cls, reg


def head_forward_L(sd, feats):
    return [F.relu(F.conv2d(f, sd[f'bbox_head.L_convs.{l}.0.weight'], sd[f'bbox_head.L_convs.{l}.0.bias'], 1, 1)) for l, f in enumerate(feats)]


# ---------------------------------------------------------------- anchors
def ssd_base_anchors(input_size=None, basesize_ratio_range=None, strides=None, ratios=None, v=V300):
    """anchor_generator.py:476-564: min/max sizes, scales [1, sqrt(max/min)], ratios [1, 1/r, r ...], scale_major=False
    (anchor_generator.py:150-193 else-branch), centre = stride/2, then index_select [0, n, 1, .., n-1]."""
    input_size, basesize_ratio_range = input_size or v.input_size, basesize_ratio_range or v.basesize_ratio_range
    strides, ratios = strides or v.STRIDES, ratios or v.RATIOS
    nl = len(strides)
    mn, mx = int(basesize_ratio_range[0] * 100), int(basesize_ratio_range[1] * 100)
    step = int(np.floor(mx - mn) / (nl - 2))
    min_sizes, max_sizes = [], []
    for r in range(mn, mx + 1, step):
        min_sizes.append(int(input_size * r / 100))
        max_sizes.append(int(input_size * (r + step) / 100))
    first = {(300, 0.15): (7, 15), (300, 0.2): (10, 20), (512, 0.1): (4, 10), (512, 0.15): (7, 15)}[(input_size, basesize_ratio_range[0])]
    min_sizes.insert(0, int(input_size * first[0] / 100))
    max_sizes.insert(0, int(input_size * first[1] / 100))
    out = []
    for k in range(nl):
        scales = torch.Tensor([1., np.sqrt(max_sizes[k] / min_sizes[k])])
        rr = [1.]
        for r in ratios[k]:
            rr += [1 / r, r]
        rr = torch.Tensor(rr)
        w = h = min_sizes[k]
        xc = yc = strides[k] / 2.
        h_r = torch.sqrt(rr)
        w_r = 1 / h_r
        ws = (w * scales[:, None] * w_r[None, :]).view(-1)
        hs = (h * scales[:, None] * h_r[None, :]).view(-1)
        base = torch.stack([xc - 0.5 * ws, yc - 0.5 * hs, xc + 0.5 * ws, yc + 0.5 * hs], dim=-1)
        ind = list(range(len(rr)))
        ind.insert(1, len(ind))
        out.append(base[torch.LongTensor(ind)])
    return out


def anchors_for(featmap_sizes, pad_shapes, v=V300):
    base = ssd_base_anchors(v=v)
    mlvl = geometry.grid_anchors(base, featmap_sizes, v.STRIDES)
    flags = [geometry.valid_flags(featmap_sizes, v.STRIDES, ps, list(v.NUM_ANCHORS)) for ps in pad_shapes]
    return mlvl, flags


def nhwc_flat(x, c):
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, c)


# ---------------------------------------------------------------- losses
def ssd_loss_single(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_total, num_classes=20, neg_pos_ratio=3,
                    beta=1.0):
    """My_L_ssd_head.py:182-215 for ONE image: cls_score [A, C+1]."""
    ce = F.cross_entropy(cls_score, labels, reduction='none') * label_weights
    pos = ((labels >= 0) & (labels < num_classes)).nonzero(as_tuple=False).reshape(-1)
    neg = (labels == num_classes).nonzero(as_tuple=False).view(-1)
    n_neg = min(neg_pos_ratio * pos.size(0), neg.size(0))
    top_neg, _ = ce[neg].topk(n_neg)
    loss_cls = (ce[pos].sum() + top_neg.sum()) / num_total
    loss_bbox = (losses.smooth_l1(bbox_pred, bbox_targets, beta) * bbox_weights).sum() / num_total
    return loss_cls[None], loss_bbox, ce


def train_step(sd, img, gt_bboxes, gt_labels, num_classes=20, v=V300):
    """SSD_L_SingleStageDetector.forward_train -> MyLSSDHead.loss -> _parse_losses."""
    B, _, H, W = img.shape
    feats = neck(sd, backbone(sd, img), v)
    cls, reg = head_forward(sd, feats)
    sizes = [tuple(f.shape[-2:]) for f in feats]
    mlvl, flags = anchors_for(sizes, [(H, W, 3)] * B, v)
    tg = geometry.get_targets(mlvl, flags, gt_bboxes, gt_labels, num_classes, assigner_cfg=ASSIGNER, coder_stds=CODER_STDS)
    n = tg['num_total_pos']
    all_cls = torch.cat([nhwc_flat(c, num_classes + 1) for c in cls], 1)
    all_reg = torch.cat([nhwc_flat(r, 4) for r in reg], 1)
    cat = {k: torch.cat(tg[k], 1) for k in ('labels', 'label_weights', 'bbox_targets', 'bbox_weights')}
    lc, lb, lnr = [], [], []
    for b in range(B):
        a, bb, c = ssd_loss_single(all_cls[b], all_reg[b], cat['labels'][b], cat['label_weights'][b], cat['bbox_targets'][b],
                                   cat['bbox_weights'][b], n, num_classes)
        lc.append(a), lb.append(bb), lnr.append(c)
    loss, _ = losses.parse_losses(dict(loss_cls=lc, loss_bbox=lb, loss_noR=lnr))
    return dict(loss=loss, loss_cls=lc, loss_bbox=lb, loss_noR=lnr, feats=feats, targets=tg, cls=cls, reg=reg, cat=cat)


def train_step_L(sd, feats, loss_noR):
    """forward_train_L -> loss_L (:302-313) -> loss_single_L (:217-224): per image 2*mean((lambda + 1e-9 - CE)^2)."""
    Ls = head_forward_L(sd, [f.detach() for f in feats])
    B = Ls[0].shape[0]
    all_L = torch.cat([nhwc_flat(l, 1).reshape(B, -1) for l in Ls], 1) + 1e-9
    ll = [torch.abs(all_L[b] - loss_noR[b].detach()).pow(2).mean() * 2 for b in range(B)]
    return dict(loss=sum(ll), loss_L=ll, Ls=Ls)


# ---------------------------------------------------------------- scoring
def pre_nms_softmax(mlvl_cls, mlvl_reg, mlvl_L, mlvl_anchors, img_shapes, scale_factors, nms_pre=1000, rescale=True):
    """My_L_ssd_head.py:325-361: scores = softmax over C+1 logits (background last), top-k on the foreground max, decode with
    stds (.1,.1,.2,.2), clip to img_shape, /scale_factor.  Same dict layout as oracle.detect.pre_nms."""
    B = mlvl_cls[0].shape[0]
    out = dict(boxes=[], scores=[], lam=[], idx=[], level_any_fg=[])
    for cls, reg, lam, anchors in zip(mlvl_cls, mlvl_reg, mlvl_L, mlvl_anchors):
        scores = cls.softmax(-1)
        out['level_any_fg'].append((scores[..., :-1].max(dim=2)[0] > 0.3).any(dim=1))          # :443-447 level gate
        A = cls.shape[1]
        anc = anchors[None].expand(B, A, 4)
        idx = torch.arange(A)[None].expand(B, A)
        if 0 < nms_pre < A:
            _, topi = detect.stable_topk(scores[..., :-1].max(-1)[0], nms_pre)
            bi = torch.arange(B).view(-1, 1).expand_as(topi)
            idx, anc, reg, scores, lam = idx[bi, topi], anc[bi, topi], reg[bi, topi], scores[bi, topi], lam[bi, topi]
        boxes = geometry.delta2bbox(anc, reg, stds=CODER_STDS, max_shape=[s[:2] for s in img_shapes])
        for k, v in zip(('boxes', 'scores', 'lam', 'idx'), (boxes, scores, lam, idx)):
            out[k].append(v)
    boxes = torch.cat(out['boxes'], dim=1)
    if rescale:
        boxes = boxes / boxes.new_tensor(np.stack(scale_factors)).unsqueeze(1)
    out['cat_boxes'], out['cat_scores'] = boxes, torch.cat(out['scores'], dim=1)
    return out


def score_images(sd, img, img_shapes=None, scale_factors=None, sampler='torch', seed=20, image_ids=None, num_classes=20,
                 uPool2='objectSum_scaleMax_classSum', heads=None, score_thr=0.02, max_per_img=200, v=V300):
    """simple_test(isEval=False, uPool='Entropy_NMS') for the SSD head (:598-..., :316-433, :435-482)."""
    B, _, H, W = img.shape
    img_shapes = img_shapes or [(H, W, 3)] * B
    scale_factors = scale_factors or [np.ones(4, np.float32)] * B
    if heads is None:
        with torch.no_grad():
            feats = neck(sd, backbone(sd, img), v)
            cls, reg = head_forward(sd, feats)
            Ls = head_forward_L(sd, feats)
    else:
        cls, reg, Ls = heads
    sizes = [tuple(c.shape[-2:]) for c in cls]
    mlvl = geometry.grid_anchors(ssd_base_anchors(v=v), sizes, v.STRIDES)
    pre = pre_nms_softmax([nhwc_flat(c, num_classes + 1) for c in cls], [nhwc_flat(r, 4) for r in reg],
                          [nhwc_flat(l, 1)[..., 0] for l in Ls], mlvl, img_shapes, scale_factors)
    dets, pos = [], []
    for b in range(B):
        d, lab, keep, inds = detect.multiclass_nms(pre['cat_boxes'][b], pre['cat_scores'][b], score_thr=score_thr, max_num=max_per_img)
        dets.append((d, lab, keep))
        pos.append(detect.get_object_idx(d, pre['cat_boxes'][b]))
    level_offsets = np.concatenate([[0], np.cumsum([a.shape[0] for a in mlvl])[:-1]])
    bins, pairs = hua.compute_obj_unc(pre, pos, sampler=sampler, seed=seed, image_ids=image_ids, level_offsets=level_offsets)
    unc = hua.aggregate_obj_scale_unc(bins, uPool2)
    return dict(unc=unc, dets=dets, pos=pos, pre=pre, bins=bins, pairs=pairs)
