"""Oracle: anchors, IoU, MaxIoU assignment, delta coder.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch fp32 on CPU.
"""
import numpy as np
import torch


# ---------------------------------------------------------------- anchors (a5)
def gen_base_anchors(strides, ratios=(0.5, 1.0, 2.0), octave_base_scale=4, scales_per_octave=3,
                     scales=None, base_sizes=None, center_offset=0.0):
    """mmdet/core/anchor/anchor_generator.py:60-112 (ctor), :150-193 (single level).

    scale_major=True: ws = w * w_ratios[:,None] * scales[None,:] -> ratio-major x scale.
    """
    if scales is None:
        octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
        scales = octave_scales * octave_base_scale
    scales = torch.Tensor(scales)
    ratios = torch.Tensor(ratios)
    if base_sizes is None:
        base_sizes = list(strides)
    out = []
    for bs in base_sizes:
        w = h = bs
        xc, yc = center_offset * w, center_offset * h
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        out.append(torch.stack([xc - 0.5 * ws, yc - 0.5 * hs, xc + 0.5 * ws, yc + 0.5 * hs], dim=-1))
    return out


def grid_anchors(base_anchors, featmap_sizes, strides):
    """anchor_generator.py:308-380: shifts (y-major, x fastest) + base anchors -> [H*W*A, 4]."""
    out = []
    for base, (fh, fw), s in zip(base_anchors, featmap_sizes, strides):
        sx = torch.arange(0, fw) * s
        sy = torch.arange(0, fh) * s
        xx = sx.repeat(len(sy))
        yy = sy.view(-1, 1).repeat(1, len(sx)).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
        out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
    return out


def valid_flags(featmap_sizes, strides, pad_shape, num_base_anchors):
    """anchor_generator.py:382-438."""
    out = []
    for (fh, fw), s, na in zip(featmap_sizes, strides, num_base_anchors):
        h, w = pad_shape[:2]
        vh = min(int(np.ceil(h / s)), fh)
        vw = min(int(np.ceil(w / s)), fw)
        vx = torch.zeros(fw, dtype=torch.bool)
        vy = torch.zeros(fh, dtype=torch.bool)
        vx[:vw] = 1
        vy[:vh] = 1
        v = vx.repeat(fh) & vy.view(-1, 1).repeat(1, fw).view(-1)
        out.append(v[:, None].expand(v.size(0), na).contiguous().view(-1))
    return out


# -------------------------------------------------------------------- IoU (a6)
def bbox_overlaps(b1, b2, eps=1e-6):
    """mmdet/core/bbox/iou_calculators/iou2d_calculator.py:212-252 (mode='iou', not aligned)."""
    rows, cols = b1.size(0), b2.size(0)
    if rows * cols == 0:
        return b1.new_zeros((rows, cols))
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[:, None] + area2[None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


def max_iou_assign(anchors, gt_bboxes, gt_labels, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                   gt_max_assign_all=True):
    """mmdet/core/bbox/assigners/max_iou_assigner.py:60-210.

    Returns assigned_gt_inds [A] int64 (-1 ignore, 0 neg, g+1 pos), max_overlaps, assigned_labels.
    """
    overlaps = bbox_overlaps(gt_bboxes, anchors)
    G, A = overlaps.shape
    assigned = overlaps.new_full((A,), -1, dtype=torch.long)
    if G == 0 or A == 0:
        max_ov = overlaps.new_zeros((A,))
        if G == 0:
            assigned[:] = 0
        labels = overlaps.new_full((A,), -1, dtype=torch.long)
        return assigned, max_ov, labels
    max_ov, argmax_ov = overlaps.max(dim=0)
    gt_max_ov, gt_argmax = overlaps.max(dim=1)
    assigned[(max_ov >= 0) & (max_ov < neg_iou_thr)] = 0
    pos = max_ov >= pos_iou_thr
    assigned[pos] = argmax_ov[pos] + 1
    for i in range(G):
        if gt_max_ov[i] >= min_pos_iou:
            if gt_max_assign_all:
                assigned[overlaps[i, :] == gt_max_ov[i]] = i + 1
            else:
                assigned[gt_argmax[i]] = i + 1
    labels = assigned.new_full((A,), -1)
    pos_inds = torch.nonzero(assigned > 0, as_tuple=False).squeeze(1)
    if pos_inds.numel() > 0:
        labels[pos_inds] = gt_labels[assigned[pos_inds] - 1]
    return assigned, max_ov, labels


# ------------------------------------------------------------------ coder (a6/a12)
def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:98-140."""
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    d = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return d.sub_(d.new_tensor(means).unsqueeze(0)).div_(d.new_tensor(stds).unsqueeze(0))


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py:144-262; rois [..., N, 4], deltas [..., N, 4];
    max_shape (H, W[, C]) or per-batch list of them."""
    d = deltas * deltas.new_tensor(stds) + deltas.new_tensor(means)
    dx, dy, dw, dh = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    x1, y1, x2, y2 = rois[..., 0], rois[..., 1], rois[..., 2], rois[..., 3]
    px = (x1 + x2) * 0.5
    py = (y1 + y2) * 0.5
    pw = x2 - x1
    ph = y2 - y1
    dxw = pw * dx
    dyh = ph * dy
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    gw = pw * dw.exp()
    gh = ph * dh.exp()
    gx = px + dxw
    gy = py + dyh
    b = torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], dim=-1)
    if max_shape is not None:
        ms = b.new_tensor(max_shape)[..., :2]
        max_xy = torch.cat([ms, ms], dim=-1).flip(-1).unsqueeze(-2)  # (W,H,W,H)
        b = torch.where(b < 0, b.new_tensor(0), b)
        b = torch.where(b > max_xy, max_xy, b)
    return b


# -------------------------------------------------------- target assembly (a6)
def get_targets(mlvl_anchors, mlvl_valid_flags_per_img, gt_bboxes_list, gt_labels_list, num_classes=20,
                assigner_cfg=None, coder_means=(0., 0., 0., 0.), coder_stds=(1., 1., 1., 1.)):
    """L_anchor_head.py:155-257 with PseudoSampler (samplers/pseudo_sampler.py:23-41),
    allowed_border=-1 (anchor/utils.py:20-46 -> inside == valid), pos_weight=-1.

    Returns per-level lists of [B, A_l(,4)] tensors + num_total_pos.
    """
    assigner_cfg = assigner_cfg or {}
    num_level_anchors = [a.size(0) for a in mlvl_anchors]
    flat_anchors = torch.cat(mlvl_anchors)
    A = flat_anchors.size(0)
    all_labels, all_lw, all_bt, all_bw, all_gt_inds = [], [], [], [], []
    num_total_pos = 0
    for flags, gtb, gtl in zip(mlvl_valid_flags_per_img, gt_bboxes_list, gt_labels_list):
        inside = torch.cat(flags)
        anchors = flat_anchors[inside]
        assigned, _, _ = max_iou_assign(anchors, gtb, gtl, **assigner_cfg)
        pos_inds = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assigned == 0, as_tuple=False).squeeze(-1).unique()
        n = anchors.size(0)
        bt = torch.zeros_like(anchors)
        bw = torch.zeros_like(anchors)
        labels = anchors.new_full((n,), num_classes, dtype=torch.long)
        lw = anchors.new_zeros(n)
        if len(pos_inds) > 0:
            pos_gt = gtb[assigned[pos_inds] - 1]
            bt[pos_inds] = bbox2delta(anchors[pos_inds], pos_gt, coder_means, coder_stds)
            bw[pos_inds] = 1.0
            labels[pos_inds] = gtl[assigned[pos_inds] - 1]
            lw[pos_inds] = 1.0
        if len(neg_inds) > 0:
            lw[neg_inds] = 1.0

        def unmap(data, fill=0):  # mmdet/core/utils/misc.py:32-42
            if data.dim() == 1:
                ret = data.new_full((A,), fill)
                ret[inside] = data
            else:
                ret = data.new_full((A,) + data.shape[1:], fill)
                ret[inside] = data
            return ret
        all_labels.append(unmap(labels, num_classes))
        all_lw.append(unmap(lw))
        all_bt.append(unmap(bt))
        all_bw.append(unmap(bw))
        gi = assigned.new_full((A,), -1)
        gi[inside] = assigned
        all_gt_inds.append(gi)
        num_total_pos += max(pos_inds.numel(), 1)

    def to_levels(t):  # anchor/utils.py:4-17
        t = torch.stack(t, 0)
        out, s = [], 0
        for n in num_level_anchors:
            out.append(t[:, s:s + n])
            s += n
        return out
    return dict(labels=to_levels(all_labels), label_weights=to_levels(all_lw),
                bbox_targets=to_levels(all_bt), bbox_weights=to_levels(all_bw),
                assigned_gt_inds=torch.stack(all_gt_inds, 0), num_total_pos=num_total_pos)
