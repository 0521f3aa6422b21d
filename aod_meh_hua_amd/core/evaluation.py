"""Detection evaluation with the fork's metric: `eval_map`, `average_precision`, `tpfp_default`, `bbox_overlaps`,
`print_map_summary` (mmdet/core/evaluation/mean_ap.py:12-57,154-239,268-405,408-472; bbox_overlaps.py:4-48) and `EvalHook`
(eval_hooks.py:9-25).

The fork rounds UP to two decimals in three places -- cumulative recall and precision curves (mean_ap.py:364-365) and every sampled
precision of the VOC07 11-point AP (:49-50) -- which shifts mAP by up to ~1 point against stock mmdet; those quirks are part of the
number the paper reports and are kept.  Host-side numpy (the metric is a sequential greedy match over a few thousand boxes, not a GPU
workload); one process instead of the reference's multiprocessing.Pool (same arithmetic, deterministic order)."""
import math

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


def average_precision(recalls, precisions, mode='area'):
    """mean_ap.py:12-57; '11points' adds ceil(prec*100)/100 per recall threshold (:49-50)."""
    no_scale = recalls.ndim == 1
    if no_scale:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    num_scales = recalls.shape[0]
    ap = np.zeros(num_scales, dtype=np.float32)
    if mode == 'area':
        zeros, ones = np.zeros((num_scales, 1), dtype=recalls.dtype), np.ones((num_scales, 1), dtype=recalls.dtype)
        mrec, mpre = np.hstack((zeros, recalls, ones)), np.hstack((zeros, precisions, zeros))
        for i in range(mpre.shape[1] - 1, 0, -1):
            mpre[:, i - 1] = np.maximum(mpre[:, i - 1], mpre[:, i])
        for i in range(num_scales):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(num_scales):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                prec = precs.max() if precs.size > 0 else 0
                ap[i] += math.ceil(prec * 100) / 100
        ap /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if no_scale else ap


def tpfp_default(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, iou_thr=0.5, area_ranges=None):
    """mean_ap.py:154-239: greedy matching in descending score order; a det whose best gt is ignored counts as neither."""
    if gt_bboxes_ignore is None:
        gt_bboxes_ignore = np.empty((0, 4), dtype=np.float32)
    gt_ignore_inds = np.concatenate((np.zeros(gt_bboxes.shape[0], dtype=bool), np.ones(gt_bboxes_ignore.shape[0], dtype=bool)))
    gt_bboxes = np.vstack((gt_bboxes, gt_bboxes_ignore))
    num_dets, num_gts = det_bboxes.shape[0], gt_bboxes.shape[0]
    if area_ranges is None:
        area_ranges = [(None, None)]
    num_scales = len(area_ranges)
    tp = np.zeros((num_scales, num_dets), dtype=np.float32)
    fp = np.zeros((num_scales, num_dets), dtype=np.float32)
    if num_gts == 0:
        if area_ranges == [(None, None)]:
            fp[...] = 1
        else:
            det_areas = (det_bboxes[:, 2] - det_bboxes[:, 0]) * (det_bboxes[:, 3] - det_bboxes[:, 1])
            for i, (min_area, max_area) in enumerate(area_ranges):
                fp[i, (det_areas >= min_area) & (det_areas < max_area)] = 1
        return tp, fp
    ious = bbox_overlaps(det_bboxes, gt_bboxes)
    ious_max, ious_argmax = ious.max(axis=1), ious.argmax(axis=1)
    sort_inds = np.argsort(-det_bboxes[:, -1])
    for k, (min_area, max_area) in enumerate(area_ranges):
        gt_covered = np.zeros(num_gts, dtype=bool)
        if min_area is None:
            gt_area_ignore = np.zeros_like(gt_ignore_inds, dtype=bool)
        else:
            gt_areas = (gt_bboxes[:, 2] - gt_bboxes[:, 0]) * (gt_bboxes[:, 3] - gt_bboxes[:, 1])
            gt_area_ignore = (gt_areas < min_area) | (gt_areas >= max_area)
        for i in sort_inds:
            if ious_max[i] >= iou_thr:
                m = ious_argmax[i]
                if not (gt_ignore_inds[m] or gt_area_ignore[m]):
                    if not gt_covered[m]:
                        gt_covered[m] = True
                        tp[k, i] = 1
                    else:
                        fp[k, i] = 1
            elif min_area is None:
                fp[k, i] = 1
            else:
                bbox = det_bboxes[i, :4]
                area = (bbox[2] - bbox[0]) * (bbox[3] - bbox[1])
                if min_area <= area < max_area:
                    fp[k, i] = 1
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
