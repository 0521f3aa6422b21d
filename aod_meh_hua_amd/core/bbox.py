"""Box-geometry plugins with the reference's registry names (mmdet/core/bbox/builder.py,
iou_calculators/builder.py): MaxIoUAssigner, PseudoSampler, DeltaXYWHBBoxCoder, BboxOverlaps2D.

On the hot path the work of assign + sample + encode for the WHOLE batch is one call into the HIP
kernels (`MaxIoUAssigner.assign_batch` -> aod_max_iou_assign); the per-object methods keep the
reference's signatures for drop-in callers and are thin tensor-op restatements (device agnostic)."""
import numpy as np
import torch

from ..mmcv_lite import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
IOU_CALCULATORS = Registry('IoU calculator')


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """iou2d_calculator.py:74-260 ('iou' / 'iof', not aligned or aligned)."""
    assert mode in ('iou', 'iof')
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if rows * cols == 0:
        return bboxes1.new_zeros(bboxes1.shape[:-2] + ((rows,) if is_aligned else (rows, cols)))
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    if is_aligned:
        lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap if mode == 'iou' else area1
    else:
        lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap if mode == 'iou' else area1[..., None]
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    def __init__(self, scale=1., dtype=None):
        self.scale, self.dtype = scale, dtype

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5] and bboxes2.size(-1) in [0, 4, 5]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """max_iou_assigner.py:11-210.  `assign_batch` is the hot path."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1, iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.gpu_assign_thr, self.match_low_quality = ignore_wrt_candidates, gpu_assign_thr, match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)
        assert match_low_quality and isinstance(neg_iou_thr, float), 'only the AL configs\' assigner modes are built'

    def assign_batch(self, flat_anchors, valid_flags, gts, gt_count, gt_labels, num_classes, means, stds):
        """All images at once on the MI355X: returns (assigned_gt_inds, labels, label_weights, bbox_targets,
        bbox_weights, num_pos) with shapes [B,A] / [B,A,4] / [B]."""
        from .. import hipops as ho
        return ho.max_iou_assign(flat_anchors, valid_flags, gts, gt_count, gt_labels, float(self.pos_iou_thr), float(self.neg_iou_thr),
                                 float(self.min_pos_iou), bool(self.gt_max_assign_all), num_classes, tuple(means), tuple(stds))

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        """Reference-signature single-image entry (one image == batch of 1 through the same kernel)."""
        dev = bboxes.device
        G = gt_bboxes.shape[0]
        gmax = max(G, 1)
        gts = torch.zeros(1, gmax, 4, device=dev)
        gts[0, :G] = gt_bboxes
        labs = torch.zeros(1, gmax, dtype=torch.long, device=dev)
        if gt_labels is not None:
            labs[0, :G] = gt_labels
        cnt = torch.tensor([G], dtype=torch.int32, device=dev)
        a, lab, _, _, _, _ = self.assign_batch(bboxes.contiguous(), None, gts, cnt, labs, -1, (0., 0., 0., 0.), (1., 1., 1., 1.))
        a = a[0]
        labels = None
        if gt_labels is not None:
            labels = torch.where(a > 0, lab[0], lab.new_full((), -1))
        return AssignResult(G, a, None, labels)


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :] if gt_bboxes.numel() else gt_bboxes.view(-1, 4)


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """samplers/pseudo_sampler.py:23-41: every assigned anchor is a sample."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, None)


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:98-140."""
    proposals, gt = proposals.float(), gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000,
               clip_border=True):
    """delta_xywh_bbox_coder.py:144-262 for [..., N, 4] deltas."""
    d = deltas * deltas.new_tensor(stds) + deltas.new_tensor(means)
    px = (rois[..., 0] + rois[..., 2]) * 0.5
    py = (rois[..., 1] + rois[..., 3]) * 0.5
    pw = rois[..., 2] - rois[..., 0]
    ph = rois[..., 3] - rois[..., 1]
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = d[..., 2].clamp(min=-max_ratio, max=max_ratio)
    dh = d[..., 3].clamp(min=-max_ratio, max=max_ratio)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * d[..., 0], py + ph * d[..., 1]
    b = torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], dim=-1)
    if clip_border and max_shape is not None:
        ms = b.new_tensor(max_shape)[..., :2]
        max_xy = torch.cat([ms, ms], dim=-1).flip(-1).unsqueeze(-2)
        b = torch.where(b < 0, b.new_tensor(0), b)
        b = torch.where(b > max_xy, max_xy, b)
    return b


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
        self.means, self.stds, self.clip_border = tuple(target_means), tuple(target_stds), clip_border
        assert not add_ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip, self.clip_border)


def bbox2result(bboxes, labels, num_classes):
    """mmdet/core/bbox/transforms.py:99-116."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
