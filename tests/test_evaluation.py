"""CPU tests of the evaluation row: aod_meh_hua_amd.core.evaluation against golden values produced by the REFERENCE's
eval_map (tests/golden/eval_map.npz, tools/golden/make_golden_eval.py) -- including the fork's ceil-to-2-decimals quirk."""
import os

import numpy as np
import torch

from aod_meh_hua_amd.core import evaluation as ev
from tests import synth

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eval_map.npz'))


def test_eval_map_matches_reference_bit_for_bit():
    for name, seed, ign in (('a', 50, True), ('b', 51, False)):
        dets, anns = synth.detection_eval_case(seed=seed, with_ignore=ign)
        for ds, tag in (('voc07', 'voc07'), (tuple(str(i) for i in range(20)), 'area')):
            m, res = ev.eval_map(dets, anns, iou_thr=0.5, dataset=ds, logger='silent')
            assert m == float(G[f'{name}_{tag}_map']), (name, tag, m)
            assert np.array_equal(np.array([r['ap'] for r in res], np.float64), G[f'{name}_{tag}_ap'])
            assert np.array_equal(np.array([r['num_gts'] for r in res]), G[f'{name}_{tag}_ngt'])
            assert np.array_equal(np.array([r['num_dets'] for r in res]), G[f'{name}_{tag}_ndet'])
            assert np.array_equal(res[3]['recall'], G[f'{name}_{tag}_rec3']) and np.array_equal(res[3]['precision'], G[f'{name}_{tag}_prec3'])
        m, res = ev.eval_map(dets, anns, scale_ranges=[(0, 64), (64, 1000)], iou_thr=0.4, dataset=None, logger='silent')
        assert np.array_equal(np.array(m, np.float64), G[f'{name}_scales_map'])
        assert np.array_equal(np.stack([r['ap'] for r in res]).astype(np.float64), G[f'{name}_scales_ap'])


def test_rounding_quirk_is_present():
    """Curves are rounded UP to two decimals (mean_ap.py:364-365) and the 11-point AP ceils every sampled precision (:49-50)."""
    dets, anns = synth.detection_eval_case(seed=50)
    _, res = ev.eval_map(dets, anns, dataset='voc07', logger='silent')
    for r in res:
        if r['recall'].size:
            assert np.allclose(r['recall'] * 100, np.round(r['recall'] * 100)) and np.allclose(r['precision'] * 100, np.round(r['precision'] * 100))
    assert float(ev.average_precision(np.array([0.1, 0.4, 0.4, 0.8, 1.0]), np.array([1.0, 0.5, 0.667, 0.5, 0.401]), '11points')) == float(G['ap_kat_11'])
    assert float(ev.average_precision(np.array([0.1, 0.4, 0.4, 0.8, 1.0]), np.array([1.0, 0.5, 0.67, 0.5, 0.4]), 'area')) == float(G['ap_kat_area'])
    # 0.401 is counted as 0.41 at recall 1.0
    assert abs(float(G['ap_kat_11']) - (1.0 * 2 + 0.67 * 3 + 0.5 * 4 + 0.41 * 2) / 11) < 1e-6


def test_tpfp_and_overlaps_edge_cases():
    empty = np.zeros((0, 4), np.float32)
    d = np.array([[0, 0, 10, 10, 0.9], [0, 0, 10, 10, 0.8], [50, 50, 60, 60, 0.7]], np.float32)
    tp, fp = ev.tpfp_default(d, empty, empty)                      # no gts: everything is a false positive
    assert tp.sum() == 0 and fp.sum() == 3
    g = np.array([[0, 0, 10, 10]], np.float32)
    tp, fp = ev.tpfp_default(d, g, empty)                          # duplicate detection of one gt: second is a false positive
    assert tp.tolist() == [[1, 0, 0]] and fp.tolist() == [[0, 1, 1]]
    tp, fp = ev.tpfp_default(d, empty, g)                          # matched to an IGNORED gt: neither tp nor fp
    assert tp.tolist() == [[0, 0, 0]] and fp.tolist() == [[0, 0, 1]]
    assert ev.bbox_overlaps(d[:, :4], empty).shape == (3, 0)
    assert np.allclose(ev.bbox_overlaps(d[:1, :4], np.array([[5, 0, 15, 10]], np.float32)), [[1 / 3]])
    assert np.allclose(ev.bbox_overlaps(d[:1, :4], np.array([[5, 0, 15, 10]], np.float32), mode='iof'), [[0.5]])


def test_bbox2result_and_voc_evaluate():
    b = torch.tensor([[0, 0, 5, 5, .9], [1, 1, 6, 6, .8], [2, 2, 7, 7, .7]])
    l = torch.tensor([2, 0, 2])
    r = ev.bbox2result(b, l, 4)
    assert [x.shape[0] for x in r] == [1, 0, 2, 0] and r[2].dtype == np.float32
    assert all(x.shape == (0, 5) for x in ev.bbox2result(torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), 3))
    dets, anns = synth.detection_eval_case(seed=50)
    out07 = ev.evaluate_voc(dets, anns, year=2007, logger='silent')
    out12 = ev.evaluate_voc(dets, anns, year=2012, logger='silent')
    assert out07['mAP'] == float(G['a_voc07_map']) and out12['mAP'] == float(G['a_area_map']) and out07['AP50'] == round(out07['mAP'], 3)


def test_vectorised_matching_equals_a_sequential_greedy_walk():
    """tpfp_default is a vectorised first-claim match; here it is checked against the literal sequential definition (walk the detections
    in descending score, a gt can be claimed once) on random crowded scenes with ignored gts, area ranges and tied scores."""
    rng = np.random.default_rng(7)
    for trial in range(60):
        nd, ng, ni = int(rng.integers(0, 40)), int(rng.integers(0, 6)), int(rng.integers(0, 3))
        def boxes(n):
            xy = rng.uniform(0, 60, (n, 2)).astype(np.float32)
            wh = rng.uniform(4, 50, (n, 2)).astype(np.float32)
            return np.hstack((xy, xy + wh)).astype(np.float32)
        sc = np.round(rng.uniform(0, 1, (nd, 1)), 1 if trial % 2 else 6).astype(np.float32)      # odd trials: many tied scores
        det, gt, ign = np.hstack((boxes(nd), sc)), boxes(ng), boxes(ni)
        for ranges in (None, [(0, 900), (900, 1e5)]):
            for thr in (0.3, 0.5):
                tp, fp = ev.tpfp_default(det, gt, ign, thr, ranges)
                allgt = np.vstack((gt, ign))
                iou = ev.bbox_overlaps(det, allgt)
                for k, (lo, hi) in enumerate(ranges or [(None, None)]):
                    taken, etp, efp = set(), np.zeros(nd), np.zeros(nd)
                    for i in np.argsort(-det[:, -1]):
                        area = (det[i, 2] - det[i, 0]) * (det[i, 3] - det[i, 1])
                        inside = lo is None or lo <= area < hi
                        if allgt.shape[0] == 0 or iou[i].max() < thr:
                            efp[i] = inside
                            continue
                        m = int(iou[i].argmax())
                        ga = (allgt[m, 2] - allgt[m, 0]) * (allgt[m, 3] - allgt[m, 1])
                        if m >= ng or (lo is not None and not lo <= ga < hi):
                            continue
                        if m in taken:
                            efp[i] = 1
                        else:
                            taken.add(m)
                            etp[i] = 1
                    assert np.array_equal(tp[k], etp) and np.array_equal(fp[k], efp), (trial, k, thr)
