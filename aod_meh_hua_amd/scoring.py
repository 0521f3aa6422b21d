"""HUA scoring pass on the HIP kernels: Lambda_L2Net._get_bboxes (Lambda_L2.py:254-384) +
ComputeObjUnc / AggregateObjScaleUnc (:489-619) for a whole batch without a host sync.

    per level   aod_softmax_rowmax  -> row max of the normalised scores, level gate
                aod_topk_stable     -> per-image top-nms_pre anchors (levels with more than nms_pre anchors)
                aod_gather_decode   -> candidates: boxes / scores(+bg) / lambda / anchor id
                (all levels in two launches: aod_pre_nms_levels)
    per batch   aod_multiclass_nms  -> dets [B,max,5], labels, keep, num_det
                aod_hua_score       -> one epistemic-uncertainty score per image

`unc` stays on the device ([B] fp32 tensor): the pool loop (apis/test.py single_gpu_uncertainty) concatenates
tensors and only syncs once per pool, instead of the reference's `.item()` per (object, level, class) bin."""
import ctypes as C
import os

import numpy as np
import torch

from . import _C
from ._C import call, ptr, stream

AGG_CODE = {'Sum': 0, 'Avg': 1, 'Max': 2}
_F4 = C.c_float * 4


def extract_agg_codes(type_str):
    """mmdet/utils/functions.py:425-436 ExtractAggFunc -> (class, scale, object) kernel codes."""
    out = {}
    for name in ('object', 'scale', 'class'):
        for part in type_str.split('_'):
            if name in part:
                out[name] = AGG_CODE[part.replace(name, '')]
    return out['class'], out.get('scale', 2), out.get('object', 0)


def nhwc_view(x, c):
    """[B, A*c, h, w] channels_last fp32 -> [B, h*w*A, c] view."""
    B = x.shape[0]
    xr = x.permute(0, 2, 3, 1)
    if not xr.is_contiguous():
        xr = xr.contiguous()
    return xr.reshape(B, -1, c)


_META = {}
_META_STATIC = None      # (hw [B,2], sc [B,4]) static device buffers of a captured scoring graph (graphs.GraphedScore refreshes them per batch)


class static_meta:
    """inside this context the image sizes / scale factors of a scoring batch are read from the given STATIC device buffers instead of
    value-keyed cached tensors: a captured scoring graph then serves every batch of its tensor shape, whatever the per-image sizes are"""

    def __init__(self, hw, sc):
        self.new = (hw, sc)

    def __enter__(self):
        global _META_STATIC
        self.prev, _META_STATIC = _META_STATIC, self.new

    def __exit__(self, *exc):
        global _META_STATIC
        _META_STATIC = self.prev
        return False


def meta_values(img_shapes, scale_factors):
    """host rows ([B,2] sizes, [B,4] scale factors) as float32 tensors: what a static-meta graph copies into its buffers"""
    hw = torch.tensor([[float(s[0]), float(s[1])] for s in img_shapes], dtype=torch.float32)
    sc = torch.tensor(np.stack([np.asarray(s, np.float32).reshape(-1)[:4] for s in scale_factors]), dtype=torch.float32)
    return hw, sc


def _meta_tensors(img_shapes, scale_factors, dev):
    """[B,2] image sizes and [B,4] scale factors on the device, cached by value: the pool loop re-uses a handful of shapes, and an
    H2D copy per batch would also break HIP-graph capture of the scoring pass."""
    if _META_STATIC is not None:
        return _META_STATIC[0], (_META_STATIC[1] if scale_factors is not None else None)
    key = (tuple((float(s[0]), float(s[1])) for s in img_shapes),
           None if scale_factors is None else tuple(tuple(float(v) for v in np.asarray(s, np.float32).reshape(-1)[:4]) for s in scale_factors), str(dev))
    hit = _META.get(key)
    if hit is None:
        if len(_META) > 256:
            _META.clear()
        hw = torch.tensor(key[0], dtype=torch.float32).to(dev)
        sc = torch.tensor(key[1], dtype=torch.float32).to(dev) if key[1] is not None else None
        hit = _META[key] = (hw, sc)
    return hit


class Candidates:
    """Outputs of the pre-NMS stage for a batch (concatenated levels)."""

    def __init__(self, boxes, scores, lam, cand_anchor, level_start, any_fg, topk_idx, rowmax=None):
        self.boxes, self.scores, self.lam, self.cand_anchor = boxes, scores, lam, cand_anchor
        self.level_start, self.any_fg, self.topk_idx = level_start, any_fg, topk_idx
        self.rowmax = rowmax              # per level [B, A_l]: max normalised class score of every anchor (before top-k)

    def max_conf(self):
        """getMaxConf (mmdet/utils/functions.py:467-476): per image, the largest class probability over all levels / anchors."""
        return torch.stack([r.amax(dim=1) for r in self.rowmax], dim=1).amax(dim=1)


def pre_nms(mlvl_cls, mlvl_reg, mlvl_L, mlvl_anchors, img_shapes, scale_factors, nms_pre, C_, means, stds, rescale=True,
            fg_thr=0.3, wh_ratio_clip=16 / 1000, normalize=True, has_bg=False):
    dev = mlvl_cls[0].device
    B = mlvl_cls[0].shape[0]
    L = len(mlvl_cls)
    cls = [nhwc_view(c.float(), C_) for c in mlvl_cls]
    reg = [nhwc_view(r.float(), 4) for r in mlvl_reg]
    lam = [nhwc_view(l.float(), 1).reshape(B, -1) for l in mlvl_L]
    A = [c.shape[1] for c in cls]
    ks = [nms_pre if 0 < nms_pre < a else a for a in A]
    n = sum(ks)
    any_fg = torch.zeros(L, B, dtype=torch.int32, device=dev)
    boxes = torch.empty(B, n, 4, device=dev)
    scores = torch.empty(B, n, C_ if has_bg else C_ + 1, device=dev)
    lam_o = torch.empty(B, n, device=dev)
    cand_anchor = torch.empty(B, n, dtype=torch.int32, device=dev)
    img_hw, sc4 = _meta_tensors(img_shapes, scale_factors if rescale else None, dev)
    level_start = [0]
    for l in range(L):
        level_start.append(level_start[-1] + ks[l])
    if L <= 8 and max(ks) <= 1024 and os.environ.get('AOD_PRE_NMS_MERGED', '1') != '0':
        # all levels in two launches (aod_pre_nms_levels): one [sum B*A] row-max buffer and one index buffer, handed out as per-level views
        rm = torch.empty(B * sum(A), device=dev)
        tk = [l for l in range(L) if ks[l] < A[l]]
        ix = torch.empty(max(B * sum(ks[l] for l in tk), 1), dtype=torch.int32, device=dev)
        anch = [a.contiguous() for a in mlvl_anchors]
        PA = C.c_void_p * L
        call('aod_pre_nms_levels', L, PA(*[ptr(t).value for t in cls]), PA(*[ptr(t).value for t in reg]), PA(*[ptr(t).value for t in lam]),
             PA(*[ptr(t).value for t in anch]), (C.c_int64 * L)(*A), (C.c_int32 * L)(*ks), B, C_, fg_thr, int(has_bg),
             2 if has_bg else int(bool(normalize)), ptr(img_hw), ptr(sc4), _F4(*means), _F4(*stds), float(wh_ratio_clip), ptr(rm), ptr(any_fg),
             ptr(ix), ptr(boxes), ptr(scores), ptr(lam_o), ptr(cand_anchor), n, stream())
        rowmaxes, idxs, r0, i0 = [], [], 0, 0
        for l in range(L):
            rowmaxes.append(rm[r0:r0 + B * A[l]].view(B, A[l]))
            r0 += B * A[l]
            if ks[l] < A[l]:
                idxs.append(ix[i0:i0 + B * ks[l]].view(B, ks[l]))
                i0 += B * ks[l]
            else:
                idxs.append(None)
        return Candidates(boxes, scores, lam_o, cand_anchor, level_start, any_fg, idxs, rowmaxes)
    c0 = a0 = 0
    idxs, rowmaxes = [], []
    for l in range(L):
        rowmax = torch.empty(B, A[l], device=dev)
        from .hipops import prof_bytes
        prof_bytes('softmax_rowmax', B * A[l] * (C_ * 4 + 4),
                   lambda: call('aod_softmax_rowmax', ptr(cls[l]), B, A[l], C_, fg_thr, ptr(rowmax), ptr(any_fg[l]), int(has_bg), stream()))
        idx = None
        if ks[l] < A[l]:
            idx = torch.empty(B, ks[l], dtype=torch.int32, device=dev)
            call('aod_topk_stable', ptr(rowmax), B, A[l], ks[l], ptr(idx), ks[l], stream())
        idxs.append(idx)
        rowmaxes.append(rowmax)
        call('aod_gather_decode', ptr(cls[l]), ptr(reg[l]), ptr(lam[l]), ptr(mlvl_anchors[l].contiguous()), ptr(idx), B, A[l], ks[l], C_,
             ks[l], ptr(img_hw), ptr(sc4), _F4(*means), _F4(*stds), float(wh_ratio_clip), ptr(boxes), ptr(scores), ptr(lam_o),
             ptr(cand_anchor), n, c0, a0, 2 if has_bg else int(bool(normalize)), stream())
        c0 += ks[l]
        a0 += A[l]
    return Candidates(boxes, scores, lam_o, cand_anchor, level_start, any_fg, idxs, rowmaxes)


def multiclass_nms_batch(boxes, scores, score_thr, iou_thr, max_num):
    B, n, C1 = scores.shape
    dev = boxes.device
    dets = torch.empty(B, max_num, 5, device=dev)
    labels = torch.empty(B, max_num, dtype=torch.int64, device=dev)
    keep = torch.empty(B, max_num, dtype=torch.int64, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(_C.lib.aod_nms_ws_bytes(B, n, C1 - 1)), 8), dtype=torch.uint8, device=dev)
    call('aod_multiclass_nms', ptr(boxes), ptr(scores), B, n, C1 - 1, float(score_thr), float(iou_thr), int(max_num), ptr(dets),
         ptr(labels), ptr(keep), ptr(num), ptr(ws), stream())
    return dets, labels, keep, num


def hua_score(cand, dets, num_det, image_ids, max_num, agg=(0, 2, 0), clsW=False, num_samples=500, seed=20, obj_score_thr=0.3,
              obj_iou_thr=0.5, fg_thr=0.3, want_pairs=False, max_pairs=None, scale_mode=False, dirichlet_cols=0):
    B, n, C1 = cand.scores.shape
    dev = cand.boxes.device
    L = len(cand.level_start) - 1
    max_pairs = max_pairs or (n if scale_mode else n * max_num)
    unc = torch.empty(B, device=dev)
    pair_count = torch.empty(B, dtype=torch.int32, device=dev)
    pair_out = torch.zeros(B, max_pairs, 4, device=dev) if want_pairs else None
    ws = torch.empty(int(_C.lib.aod_hua_ws_bytes(B, max_pairs)), dtype=torch.uint8, device=dev)
    call('aod_hua_score', ptr(cand.boxes), ptr(cand.scores), ptr(cand.lam), ptr(cand.cand_anchor), ptr(dets), ptr(num_det),
         (C.c_int32 * (L + 1))(*cand.level_start), ptr(cand.any_fg), ptr(image_ids), B, n, L, C1 - 1, int(max_num), float(obj_score_thr),
         float(obj_iou_thr), float(fg_thr), int(num_samples), int(seed), (C.c_int32 * 3)(*agg), int(bool(clsW)), int(bool(scale_mode)),
         int(dirichlet_cols), ptr(unc), ptr(pair_out),
         int(max_pairs), ptr(pair_count), ptr(ws), stream())
    return (unc, pair_count, pair_out) if want_pairs else unc


def score_batch(head, mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, cfg, rescale=False, with_nms=True,
                **kwargs):
    """Body of Lambda_L2Net._get_bboxes for `last_activation == 'relu'`.

    isEval=True (detection for mAP)  -> list of (det_bboxes [k,5], det_labels [k]) per image.
    isUnc with uPool == 'Entropy_NMS' -> (det_results, unc [B] device tensor)."""
    assert head.last_activation in ('relu', 'softmax')
    has_bg = head.last_activation == 'softmax'         # SSD: 21 logits incl. background (My_L_ssd_head.py:331-345)
    C_ = head.cls_out_channels
    na = head.num_anchors if isinstance(head.num_anchors, (list, tuple)) else [head.num_anchors] * len(mlvl_cls_scores)
    isUnc = kwargs.get('isUnc')
    uPool = kwargs.get('uPool')
    if isUnc and uPool == 'Entropy_NoNMS':
        raise NotImplementedError('uncertainty_pool=Entropy_NoNMS crashes in the reference too (ComputeScaleUnc with L_scores=None)')
    if isUnc and uPool == 'Entropy_ALL':
        # Lambda_L2.py:281-283 (no top-k), :354 (no NMS), :364-365 ComputeScaleUnc + AggregateScaleUnc
        assert not has_bg, 'Entropy_ALL is built for the RetinaNet evidence head'
        cand = pre_nms(mlvl_cls_scores, mlvl_bbox_preds, kwargs['L_scores'], mlvl_anchors, img_shapes, scale_factors, -1, C_,
                       head.bbox_coder.means, head.bbox_coder.stds, rescale=rescale, normalize=False)
        B = cand.boxes.shape[0]
        image_ids = kwargs.get('image_ids')
        if image_ids is None:
            image_ids = torch.arange(B, device=cand.boxes.device, dtype=torch.int64) + int(kwargs.get('batchIdx', 0)) * B
        cls_code, scale_code, _ = extract_agg_codes(kwargs['uPool2'] if 'object' in kwargs['uPool2'] else 'objectSum_' + kwargs['uPool2'])
        unc = hua_score(cand, None, None, image_ids.to(torch.int64).contiguous(), 1, (cls_code, scale_code, 0), False,
                        seed=kwargs.get('hua_seed', 20), scale_mode=True)
        det_results = [(cand.boxes[b], cand.scores[b]) for b in range(B)]
        if kwargs.get('_return_internals'):
            return det_results, unc, dict(cand=cand)
        return det_results, unc
    L_scores = kwargs.get('L_scores')
    if L_scores is None:   # plain detection: lambda is not needed, reuse zeros
        L_scores = [torch.zeros(c.shape[0], a, c.shape[2], c.shape[3], device=c.device).contiguous(memory_format=torch.channels_last)
                    for c, a in zip(mlvl_cls_scores, na)]
    nms_pre = cfg.get('nms_pre', -1)
    cand = pre_nms(mlvl_cls_scores, mlvl_bbox_preds, L_scores, mlvl_anchors, img_shapes, scale_factors, nms_pre, C_,
                   head.bbox_coder.means, head.bbox_coder.stds, rescale=rescale, has_bg=has_bg)
    if not with_nms:
        return [(cand.boxes[b], cand.scores[b]) for b in range(cand.boxes.shape[0])]
    max_num = cfg.max_per_img
    dets, labels, keep, num = multiclass_nms_batch(cand.boxes, cand.scores, cfg.score_thr, cfg.nms.get('iou_threshold', 0.5), max_num)
    B = dets.shape[0]
    if not isUnc or kwargs.get('isEval'):
        nh = num.cpu().tolist()           # evaluation path: variable-length results are part of the interface
        return [(dets[b, :nh[b]], labels[b, :nh[b]]) for b in range(B)]
    image_ids = kwargs.get('image_ids')
    if image_ids is None:
        bs = kwargs.get('batchIdx', 0)
        image_ids = torch.arange(B, device=dets.device, dtype=torch.int64) + int(bs) * B
    agg = extract_agg_codes(kwargs['uPool2'])
    unc = hua_score(cand, dets, num, image_ids.to(torch.int64).contiguous(), max_num, agg, kwargs.get('clsW', False),
                    seed=kwargs.get('hua_seed', 20), dirichlet_cols=C_ if has_bg else 0)
    det_results = [(dets[b], labels[b]) for b in range(B)]   # zero-padded to max_per_img rows (num_det rows are valid)
    if kwargs.get('_return_internals'):
        return det_results, unc, dict(cand=cand, dets=dets, labels=labels, keep=keep, num=num)
    if kwargs.get('saveMaxConf'):            # Lambda_L2.py:375-380: third output = per-image max confidence (device tensor here)
        return det_results, unc, cand.max_conf()
    return det_results, unc
