"""Oracle: scoring pre-NMS pipeline, multiclass NMS, object binning.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch

from .geometry import bbox_overlaps, delta2bbox


def stable_topk(x, k):
    """Row-wise top-k, ties broken by LOWER index first.  torch.topk's tie order is
    implementation-defined (Lambda_L2.py:290); the build pins stable order."""
    idx = torch.sort(x, dim=-1, descending=True, stable=True)[1][..., :k]
    return torch.gather(x, -1, idx), idx


def pre_nms(mlvl_cls, mlvl_reg, mlvl_L, mlvl_anchors, img_shapes, scale_factors, nms_pre=1000,
            num_classes=20, rescale=True):
    """Lambda_L2Net._get_bboxes part 1 (Lambda_L2.py:264-326) on NHWC-flattened inputs:
    mlvl_cls[l] [B, A_l, C], mlvl_reg[l] [B, A_l, 4], mlvl_L[l] [B, A_l].

    alphas = softmax; scores = alphas / (sum(alphas) + 1e-20 + 1e-9); per-level top-k of
    row max; gather; decode (clip to img_shape); cat; /scale_factor; append bg column."""
    B = mlvl_cls[0].shape[0]
    out = dict(boxes=[], scores=[], alphas=[], lam=[], idx=[], level_any_fg=[], rowmax=[])
    for cls, reg, lam, anchors in zip(mlvl_cls, mlvl_reg, mlvl_L, mlvl_anchors):
        alphas = cls.softmax(dim=2)
        S = alphas.sum(dim=2, keepdim=True) + 1e-20
        scores = alphas / (S + 1e-9)
        # Lambda_L2.py:497-501 level gate uses the un-normalised softmax max over ALL anchors
        out['level_any_fg'].append((alphas.max(dim=2)[0] > 0.3).any(dim=1))
        out['rowmax'].append(scores.max(-1)[0])            # every anchor of the level, before top-k
        A = cls.shape[1]
        anc = anchors[None].expand(B, A, 4)
        idx = torch.arange(A)[None].expand(B, A)
        if 0 < nms_pre < A:
            _, topi = stable_topk(scores.max(-1)[0], nms_pre)
            bi = torch.arange(B).view(-1, 1).expand_as(topi)
            idx, anc, reg, scores, alphas, lam = idx[bi, topi], anc[bi, topi], reg[bi, topi], \
                scores[bi, topi], alphas[bi, topi], lam[bi, topi]
        boxes = delta2bbox(anc, reg, max_shape=[s[:2] for s in img_shapes])
        for k, v in zip(('boxes', 'scores', 'alphas', 'lam', 'idx'), (boxes, scores, alphas, lam, idx)):
            out[k].append(v)
    boxes = torch.cat(out['boxes'], dim=1)
    if rescale:
        boxes = boxes / boxes.new_tensor(np.stack(scale_factors)).unsqueeze(1)
    scores = torch.cat(out['scores'], dim=1)
    scores = torch.cat([scores, scores.new_zeros(B, scores.shape[1], 1)], dim=-1)
    out['cat_boxes'], out['cat_scores'] = boxes, scores
    return out


def nms_cpu(boxes, scores, iou_thr):
    """mmcv-full 1.3.8 nms_cpu (NOT in /root/reference; call site core/post_processing/bbox_nms.py:84):
    score-descending (stable) greedy, offset 0, suppress when inter/(a_i+a_j-inter) > thr."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    b = boxes.float().numpy()
    order = torch.sort(scores.float(), dim=0, descending=True, stable=True)[1].numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    sup = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thr)
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(invalid='ignore', divide='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        sup[rest[ovr > thr]] = True
    return torch.as_tensor(np.array(keep, dtype=np.int64))


def multiclass_nms(boxes, scores_with_bg, score_thr=0.05, iou_thr=0.5, max_num=100):
    """mmdet/core/post_processing/bbox_nms.py:7-93 + mmcv batched_nms (class offset
    boxes + label*(max+1); >= 10000 candidates -> per-class path re-sorted by score).
    Returns dets [k,5], labels [k], keep [k] (indices into the thresholded list),
    and `inds` (thresholded list -> flat (candidate*C + class))."""
    C = scores_with_bg.size(1) - 1
    n = scores_with_bg.size(0)
    b = boxes[:, None].expand(n, C, 4).reshape(-1, 4)
    s = scores_with_bg[:, :-1].reshape(-1)
    lab = torch.arange(C)[None].expand(n, C).reshape(-1)
    inds = (s > score_thr).nonzero(as_tuple=False).squeeze(1)
    b, s, lab = b[inds], s[inds], lab[inds]
    if b.numel() == 0:
        return torch.cat([b, s[:, None]], -1), lab, torch.zeros(0, dtype=torch.long), inds
    off = lab.to(b) * (b.max() + 1)
    bn = b + off[:, None]
    if bn.shape[0] < 10000:
        keep = nms_cpu(bn, s, iou_thr)
    else:
        mask = torch.zeros_like(s, dtype=torch.bool)
        for c in torch.unique(lab):
            m = (lab == c).nonzero(as_tuple=False).view(-1)
            mask[m[nms_cpu(bn[m], s[m], iou_thr)]] = True
        keep = mask.nonzero(as_tuple=False).view(-1)
        keep = keep[torch.sort(s[keep], descending=True, stable=True)[1]]
    if max_num > 0:
        keep = keep[:max_num]
    dets = torch.cat([b[keep], s[keep][:, None]], -1)
    return dets, lab[keep], keep, inds


def get_object_idx(det_bbox, cand_boxes, score_thr=0.3, iou_thr=0.5):
    """Lambda_L2.py:343-349 GetObjectIdx."""
    filt = det_bbox[det_bbox[:, -1] > score_thr]
    return bbox_overlaps(cand_boxes, filt[:, :4]) > iou_thr
