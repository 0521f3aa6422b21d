"""Detection evaluation with the fork's metric: `eval_map`, `average_precision`, `tpfp_default`, `bbox_overlaps`,
`print_map_summary` (mmdet/core/evaluation/mean_ap.py:12-57,154-239,268-405,408-472; bbox_overlaps.py:4-48) and `EvalHook`
(eval_hooks.py:9-25).

The fork rounds UP to two decimals in three places -- cumulative recall and precision curves (mean_ap.py:364-365) and every sampled
precision of the VOC07 11-point AP (:49-50) -- which shifts mAP by up to ~1 point against stock mmdet; those quirks are part of the
number the paper reports and are kept.  Host-side numpy (the metric is a sequential greedy match over a few thousand boxes, not a GPU
workload); one process instead of the reference's multiprocessing.Pool (same arithmetic, deterministic order)."""
import numpy as np


def bbox_overlaps(bboxes1, bboxes2, mode='iou', eps=1e-6):
    """bbox_overlaps.py:4-48 (float32; the reference's row loop over the smaller set is an outer broadcast here: same values)."""
    assert mode in ('iou', 'iof')
    b1, b2 = bboxes1.astype(np.float32), bboxes2.astype(np.float32)
    rows, cols = b1.shape[0], b2.shape[0]
    if rows * cols == 0:
        return np.zeros((rows, cols), dtype=np.float32)
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    xs = np.maximum(b1[:, None, 0], b2[None, :, 0])
    ys = np.maximum(b1[:, None, 1], b2[None, :, 1])
    xe = np.minimum(b1[:, None, 2], b2[None, :, 2])
    ye = np.minimum(b1[:, None, 3], b2[None, :, 3])
    overlap = np.maximum(xe - xs, 0) * np.maximum(ye - ys, 0)
    union = area1[:, None] + area2[None, :] - overlap if mode == 'iou' else np.broadcast_to(area1[:, None], overlap.shape)
    return (overlap / np.maximum(union, np.float32(eps))).astype(np.float32)


def _running_max_from_the_right(p):
    return np.maximum.accumulate(p[..., ::-1], axis=-1)[..., ::-1]


def average_precision(recalls, precisions, mode='area'):
    """Behaviour of mean_ap.py:12-57.  'area': area under the monotone precision envelope, summed over the recall steps; '11points':
    mean over recall thresholds 0, 0.1, ..., 1 of the best precision at recall >= threshold, each sample rounded UP to two decimals
    (the fork's quirk, :49-50).  One curve ([n]) or one per scale ([scales, n]); float32 result like the reference's accumulator."""
    single = recalls.ndim == 1
    rec, prec = np.atleast_2d(recalls), np.atleast_2d(precisions)
    assert rec.shape == prec.shape and rec.ndim == 2
    n_curves = rec.shape[0]
    out = np.zeros(n_curves, dtype=np.float32)
    if mode == 'area':
        pad0, pad1 = np.zeros((n_curves, 1), rec.dtype), np.ones((n_curves, 1), rec.dtype)
        r = np.concatenate((pad0, rec, pad1), axis=1)                         # recall axis closed at 0 and 1
        envelope = _running_max_from_the_right(np.concatenate((pad0, prec, pad0), axis=1))
        width = r[:, 1:] - r[:, :-1]
        for c in range(n_curves):
            step = r[c, 1:] != r[c, :-1]                                       # only the columns where recall moves contribute
            out[c] = np.sum(width[c, step] * envelope[c, 1:][step])
    elif mode == '11points':
        thresholds = np.arange(0, 1 + 1e-3, 0.1)
        reached = rec[:, None, :] >= thresholds[None, :, None]                # [curve, threshold, point]
        best = np.where(reached, prec[:, None, :], -np.inf).max(axis=2, initial=-np.inf)
        best = np.where(reached.any(axis=2), best, 0).astype(prec.dtype)      # no point reaches the threshold: 0
        samples = np.ceil(best * 100).astype(np.float64) / 100                 # rounded UP, in the curve's own dtype
        for c in range(n_curves):
            acc = np.float32(0)
            for v in samples[c]:                                               # float32 running sum, threshold by threshold
                acc = np.float32(acc + np.float32(v))
            out[c] = acc
        out /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return out[0] if single else out


def _box_areas(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def tpfp_default(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, iou_thr=0.5, area_ranges=None):
    """Behaviour of mean_ap.py:154-239, vectorised.  Every detection points at its best-IoU gt; among the detections that reach
    `iou_thr` on a countable gt, the highest-scoring one is the true positive and the others are false positives; a detection whose
    best gt is an ignored one (or outside the area range) is neither; a detection below the threshold is a false positive when its own
    area lies in the range.  Returns (tp, fp), each [scales, num_dets] float32 in the detections' input order."""
    n_real = gt_bboxes.shape[0]
    if gt_bboxes_ignore is not None and gt_bboxes_ignore.shape[0]:
        gt_bboxes = np.vstack((gt_bboxes, gt_bboxes_ignore))
    n_det, n_gt = det_bboxes.shape[0], gt_bboxes.shape[0]
    flagged_ignore = np.arange(n_gt) >= n_real
    ranges = [(None, None)] if area_ranges is None else list(area_ranges)
    tp = np.zeros((len(ranges), n_det), dtype=np.float32)
    fp = np.zeros((len(ranges), n_det), dtype=np.float32)
    det_area = _box_areas(det_bboxes[:, :4]) if n_det else np.zeros(0, det_bboxes.dtype)

    def det_in_range(lo, hi):
        return np.ones(n_det, bool) if lo is None else (det_area >= lo) & (det_area < hi)

    if n_gt == 0:
        for k, (lo, hi) in enumerate(ranges):
            fp[k, det_in_range(lo, hi)] = 1
        return tp, fp
    iou = bbox_overlaps(det_bboxes, gt_bboxes)
    best_gt, best_iou = iou.argmax(axis=1), iou.max(axis=1)
    reaches = best_iou >= iou_thr
    order = np.argsort(-det_bboxes[:, -1])                                    # the reference's tie order is this call's
    rank = np.empty(n_det, dtype=np.int64)
    rank[order] = np.arange(n_det)
    gt_area = _box_areas(gt_bboxes)
    for k, (lo, hi) in enumerate(ranges):
        uncounted = flagged_ignore if lo is None else flagged_ignore | (gt_area < lo) | (gt_area >= hi)
        claim = np.flatnonzero(reaches & ~uncounted[best_gt])                  # detections competing for a countable gt
        first = np.full(n_gt, n_det, dtype=np.int64)
        np.minimum.at(first, best_gt[claim], rank[claim])                      # best-ranked claimant per gt
        won = rank[claim] == first[best_gt[claim]]
        tp[k, claim[won]] = 1
        fp[k, claim[~won]] = 1
        fp[k, ~reaches & det_in_range(lo, hi)] = 1
    return tp, fp


def get_cls_results(det_results, annotations, class_id):
    """mean_ap.py:242-265."""
    cls_dets = [img_res[class_id] for img_res in det_results]
    cls_gts, cls_gts_ignore = [], []
    for ann in annotations:
        cls_gts.append(ann['bboxes'][ann['labels'] == class_id, :])
        if ann.get('labels_ignore', None) is not None:
            cls_gts_ignore.append(ann['bboxes_ignore'][ann['labels_ignore'] == class_id, :])
        else:
            cls_gts_ignore.append(np.empty((0, 4), dtype=np.float32))
    return cls_dets, cls_gts, cls_gts_ignore


def eval_map(det_results, annotations, scale_ranges=None, iou_thr=0.5, dataset=None, logger=None, tpfp_fn=None, nproc=4):
    """mean_ap.py:268-405.  Returns (mean_ap, [dict(num_gts, num_dets, recall, precision, ap) per class])."""
    assert len(det_results) == len(annotations)
    num_scales = len(scale_ranges) if scale_ranges is not None else 1
    num_classes = len(det_results[0])
    area_ranges = [(rg[0] ** 2, rg[1] ** 2) for rg in scale_ranges] if scale_ranges is not None else None
    tpfp_fn = tpfp_fn or tpfp_default
    eval_results = []
    for c in range(num_classes):
        cls_dets, cls_gts, cls_gts_ignore = get_cls_results(det_results, annotations, c)
        tpfp = [tpfp_fn(d, g, gi, iou_thr, area_ranges) for d, g, gi in zip(cls_dets, cls_gts, cls_gts_ignore)]
        tp, fp = tuple(zip(*tpfp))
        num_gts = np.zeros(num_scales, dtype=int)
        for bbox in cls_gts:
            if area_ranges is None:
                num_gts[0] += bbox.shape[0]
            else:
                gt_areas = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
                for k, (min_area, max_area) in enumerate(area_ranges):
                    num_gts[k] += np.sum((gt_areas >= min_area) & (gt_areas < max_area))
        cls_dets = np.vstack(cls_dets)
        num_dets = cls_dets.shape[0]
        sort_inds = np.argsort(-cls_dets[:, -1])
        tp = np.cumsum(np.hstack(tp)[:, sort_inds], axis=1)
        fp = np.cumsum(np.hstack(fp)[:, sort_inds], axis=1)
        eps = np.finfo(np.float32).eps
        recalls = np.ceil(tp / np.maximum(num_gts[:, np.newaxis], eps) * 100) / 100          # the fork's ceil-to-2-decimals (:364-365)
        precisions = np.ceil(tp / np.maximum((tp + fp), eps) * 100) / 100
        if scale_ranges is None:
            recalls, precisions, num_gts = recalls[0, :], precisions[0, :], num_gts.item()
        ap = average_precision(recalls, precisions, 'area' if dataset != 'voc07' else '11points')
        eval_results.append(dict(num_gts=num_gts, num_dets=num_dets, recall=recalls, precision=precisions, ap=ap))
    if scale_ranges is not None:
        all_ap = np.vstack([r['ap'] for r in eval_results])
        all_num_gts = np.vstack([r['num_gts'] for r in eval_results])
        mean_ap = [all_ap[all_num_gts[:, i] > 0, i].mean() if np.any(all_num_gts[:, i] > 0) else 0.0 for i in range(num_scales)]
    else:
        aps = [r['ap'] for r in eval_results if r['num_gts'] > 0]
        mean_ap = np.array(aps).mean().item() if aps else 0.0
    print_map_summary(mean_ap, eval_results, dataset, area_ranges, logger=logger)
    return mean_ap, eval_results


def print_map_summary(mean_ap, results, dataset=None, scale_ranges=None, logger=None):
    """mean_ap.py:408-472 (tabulate instead of terminaltables)."""
    if logger == 'silent':
        return
    from ..mmcv_lite import print_log
    if isinstance(results[0]['ap'], np.ndarray):
        num_scales = len(results[0]['ap'])
    else:
        num_scales = 1
    if scale_ranges is not None:
        assert len(scale_ranges) == num_scales
    num_classes = len(results)
    recalls = np.zeros((num_scales, num_classes), dtype=np.float32)
    aps = np.zeros((num_scales, num_classes), dtype=np.float32)
    num_gts = np.zeros((num_scales, num_classes), dtype=int)
    for i, cls_result in enumerate(results):
        if cls_result['recall'].size > 0:
            recalls[:, i] = np.array(cls_result['recall'], ndmin=2)[:, -1]
        aps[:, i] = cls_result['ap']
        num_gts[:, i] = cls_result['num_gts']
    if dataset is None or isinstance(dataset, str):
        label_names = [str(i) for i in range(num_classes)] if dataset != 'voc07' else list(VOC_CLASSES)
    else:
        label_names = list(dataset)
    mean_ap = mean_ap if isinstance(mean_ap, list) else [mean_ap]
    for i in range(num_scales):
        if scale_ranges is not None:
            print_log(f'Scale range {scale_ranges[i]}', logger=logger)
        rows = [[label_names[j], num_gts[i, j], results[j]['num_dets'], f'{recalls[i, j]:.3f}', f'{aps[i, j]:.3f}'] for j in range(num_classes)]
        rows.append(['mAP', '', '', '', f'{mean_ap[i]:.3f}'])
        try:
            from tabulate import tabulate
            table = tabulate(rows, headers=['class', 'gts', 'dets', 'recall', 'ap'], tablefmt='grid')
        except ImportError:
            table = '\n'.join(' | '.join(str(c) for c in r) for r in rows)
        print_log('\n' + table, logger=logger)


VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse',
               'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


def bbox2result(bboxes, labels, num_classes):
    """mmdet/core/bbox/transforms.py:99-116: (n,5) dets + (n,) labels -> list[num_classes] of (k,5) float32 arrays."""
    import torch
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def evaluate_voc(results, annotations, year=2007, classes=VOC_CLASSES, metric='mAP', logger=None, iou_thr=0.5):
    """VOCDataset.evaluate (datasets/voc.py:37-94), metric='mAP': VOC07 -> 11-point AP, else area under the (rounded) curve."""
    if not isinstance(metric, str):
        assert len(metric) == 1
        metric = metric[0]
    if metric != 'mAP':
        raise KeyError(f'metric {metric} is not supported')
    iou_thrs = [iou_thr] if isinstance(iou_thr, float) else list(iou_thr)
    ds_name = 'voc07' if year == 2007 else classes
    out, mean_aps = {}, []
    for thr in iou_thrs:
        mean_ap, _ = eval_map(results, annotations, scale_ranges=None, iou_thr=thr, dataset=ds_name, logger=logger)
        mean_aps.append(mean_ap)
        out[f'AP{int(thr * 100):02d}'] = round(mean_ap, 3)
    out['mAP'] = sum(mean_aps) / len(mean_aps)
    return out
