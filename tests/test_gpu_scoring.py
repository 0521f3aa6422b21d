"""GPU parity of the scoring pass (softmax/top-k/decode/NMS/HUA kernels through the C ABI) against
(a) the golden artifacts the REFERENCE produced for the planted-logit inputs (tests/golden/scoring.npz) and
(b) the CPU oracle, including a value-by-value check of the Philox Dirichlet sampler."""
import os

import numpy as np
import pytest
import torch

from oracle import detect as odetect
from oracle import hua as ohua
from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.fixture(scope='module')
def run():
    from aod_meh_hua_amd import scoring
    from aod_meh_hua_amd.core.anchor import AnchorGenerator
    from aod_meh_hua_amd.core.bbox import DeltaXYWHBBoxCoder

    class Head:
        last_activation, cls_out_channels, num_anchors = 'relu', 20, 9
        bbox_coder = DeltaXYWHBBoxCoder()
    cls_p, reg_p, L_p = synth.planted_heads(2, 128, 128)
    mt = synth.metas(2, 128, 128, scale=1.25)
    ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    sizes = [tuple(c.shape[-2:]) for c in cls_p]
    anchors = ag.grid_anchors(sizes, 'cuda')
    cfg = Cfg(nms_pre=1000, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
    det, unc, internals = scoring.score_batch(Head(), [c.cuda() for c in cls_p], [r.cuda() for r in reg_p], anchors,
                                              [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt], cfg, rescale=True, with_nms=True,
                                              isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', isEval=False,
                                              L_scores=[l.cuda() for l in L_p], _return_internals=True, batchIdx=0)
    torch.cuda.synchronize()
    # CPU oracle on the same inputs
    o = omodel.score_images(None, torch.zeros(2, 3, 128, 128), [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt],
                            sampler='philox', seed=20, heads=(cls_p, reg_p, L_p))
    return dict(unc=unc, it=internals, o=o, gold=np.load(os.path.join(G, 'scoring.npz')), scoring=scoring)


def test_topk_indices_and_candidates(run):
    it, g, o = run['it'], run['gold'], run['o']
    cand = it['cand']
    assert cand.level_start == [0, 1000, 1576, 1720, 1756, 1765]
    idx0 = cand.topk_idx[0].cpu().numpy()
    assert np.array_equal(idx0, g['topk_idx'][:, :1000])                      # exact top-k order vs the reference
    assert all(i is None for i in cand.topk_idx[1:])
    assert np.allclose(cand.boxes.cpu().numpy(), g['boxes_cat'], rtol=1e-5, atol=1e-4)
    assert np.allclose(cand.scores.cpu().numpy(), o['pre']['cat_scores'].numpy(), rtol=1e-5, atol=1e-8)
    assert np.array_equal(cand.lam.cpu().numpy(), g['lam'])
    assert bool(cand.any_fg.cpu().bool().all()) == bool(torch.stack([torch.as_tensor(x) for x in o['pre']['level_any_fg']]).all())


def test_nms_keep_labels_exact(run):
    it, g = run['it'], run['gold']
    num = it['num'].cpu().tolist()
    for b in range(2):
        gd = g[f'det{b}']
        assert num[b] == gd.shape[0]
        assert np.array_equal(it['keep'][b, :num[b]].cpu().numpy(), g[f'keep{b}'])
        assert np.array_equal(it['labels'][b, :num[b]].cpu().numpy(), gd[:, 5].astype(np.int64))
        assert np.allclose(it['dets'][b, :num[b]].cpu().numpy(), gd[:, :5], rtol=1e-5, atol=1e-4)
        assert (it['keep'][b, num[b]:] == -1).all()


def test_nms_matches_oracle_on_same_candidates(run):
    """NMS kernel vs the oracle restatement fed with the KERNEL's own boxes/scores (bit-exact contract)."""
    it = run['it']
    cand = it['cand']
    for b in range(2):
        d, lab, keep, inds = odetect.multiclass_nms(cand.boxes[b].cpu(), cand.scores[b].cpu())
        n = int(it['num'][b])
        assert n == len(keep)
        assert np.array_equal(it['keep'][b, :n].cpu().numpy(), keep.numpy())
        assert torch.equal(it['dets'][b, :n].cpu(), d)


def test_hua_pairs_and_values_vs_philox_oracle(run):
    sc, it, o = run['scoring'], run['it'], run['o']
    cand = it['cand']
    ids = torch.arange(2, device='cuda', dtype=torch.int64)
    unc, pc, pout = sc.hua_score(cand, it['dets'], it['num'], ids, 100, want_pairs=True, seed=20)
    torch.cuda.synchronize()
    pc = pc.cpu().tolist()
    pout = pout.cpu().numpy()
    # oracle pairs, in (level, image) blocks -> regroup per image in level order
    for b in range(2):
        exp = [p for p in o['pairs'] if p['image'] == b]
        exp.sort(key=lambda p: p['level'])
        lvl_off = np.cumsum([0] + [1000, 576, 144, 36, 9])
        ec = np.concatenate([p['cand'].numpy() + lvl_off[p['level']] for p in exp])
        eo = np.concatenate([p['obj'].numpy() for p in exp])
        ee = np.concatenate([p['epi'].numpy() for p in exp])
        assert pc[b] == len(ec) and pc[b] > 50
        got = pout[b, :pc[b]]
        assert np.array_equal(got[:, 0].astype(np.int64), ec) and np.array_equal(got[:, 1].astype(np.int64), eo)   # pair order = nonzero()
        # same counter-based RNG stream: per-pair epistemic agrees up to transcendental ulps / rare accept flips
        err = np.abs(got[:, 3] - ee)
        print('pair epi err: median', np.median(err), 'p99', np.percentile(err, 99), 'max', err.max())
        assert np.median(err) < 2e-5 and (err < 5e-3).all(), (np.median(err), err.max())
    ounc = np.array(o['unc'])
    assert np.allclose(unc.cpu().numpy(), ounc, rtol=2e-3), (unc, ounc)
    assert np.allclose(run['unc'].cpu().numpy(), unc.cpu().numpy(), rtol=0, atol=0)      # deterministic


def test_hua_statistics_vs_reference_mc(run):
    """Image scores vs the reference's 20 reseeded MC-500 runs (4 sigma + 2 %), several seeds."""
    sc, it, g = run['scoring'], run['it'], run['gold']
    mu, sd = g['unc_runs'].mean(0), g['unc_runs'].std(0)
    ids = torch.arange(2, device='cuda', dtype=torch.int64)
    vals = []
    for seed in (1, 2, 3, 20):
        u = sc.hua_score(it['cand'], it['dets'], it['num'], ids, 100, seed=seed).cpu().numpy()
        assert (np.abs(u - mu) <= 4 * sd + 0.02 * mu).all(), (seed, u, mu, sd)
        vals.append(u)
    assert np.std(np.stack(vals), 0).max() > 0          # the seed matters
    # partition invariance: each image scored alone (as another rank would) gives the same bits
    for b in range(2):
        c = it['cand']
        sub = sc.Candidates(c.boxes[b:b + 1].contiguous(), c.scores[b:b + 1].contiguous(), c.lam[b:b + 1].contiguous(),
                            c.cand_anchor[b:b + 1].contiguous(), c.level_start, c.any_fg[:, b:b + 1].contiguous(), None)
        u1 = sc.hua_score(sub, it['dets'][b:b + 1].contiguous(), it['num'][b:b + 1].contiguous(), ids[b:b + 1].contiguous(), 100, seed=20)
        assert float(u1[0]) == float(run['unc'][b])


def _bins_from_kernel_pairs(cand, pc, pout, nobj):
    """(object, level, class) bins rebuilt on the host from the kernel's own per-pair epistemic values."""
    B = pout.shape[0]
    ls = cand.level_start
    bins = []
    sc = cand.scores.cpu()
    for b in range(B):
        img = [[{} for _ in range(len(ls) - 1)] for _ in range(nobj[b])]
        acc = {}
        for k in range(pc[b]):
            c, o, epi = int(pout[b, k, 0]), int(pout[b, k, 1]), float(pout[b, k, 3])
            lvl = max(l for l in range(len(ls) - 1) if c >= ls[l])
            cls = int(sc[b, c, :-1].argmax())
            acc.setdefault((o, lvl, cls), []).append(epi)
        for (o, lvl, cls), v in acc.items():
            img[o][lvl][cls] = float(np.mean(np.asarray(v, np.float32), dtype=np.float32))
        bins.append(img)
    return bins


def test_aggregation_modes_and_empty(run):
    sc, it, o = run['scoring'], run['it'], run['o']
    ids = torch.arange(2, device='cuda', dtype=torch.int64)
    _, pc, pout = sc.hua_score(it['cand'], it['dets'], it['num'], ids, 100, want_pairs=True, seed=20)
    nobj = [int((it['dets'][b, :int(it['num'][b]), 4] > 0.3).sum()) for b in range(2)]
    bins = _bins_from_kernel_pairs(it['cand'], pc.cpu().tolist(), pout.cpu().numpy(), nobj)
    obins = o['bins']          # oracle bins from the same Philox stream (score_images passes the level offsets)
    for mode in ('objectAvg_scaleAvg_classAvg', 'objectMax_scaleSum_classMax', 'objectSum_scaleMax_classSum', 'objectSum_scaleAvg_classMax'):
        u = sc.hua_score(it['cand'], it['dets'], it['num'], ids, 100, agg=sc.extract_agg_codes(mode), seed=20).cpu().numpy()
        assert np.allclose(u, ohua.aggregate_obj_scale_unc(bins, mode), rtol=1e-5), mode          # aggregation logic, exact inputs
        assert np.allclose(u, ohua.aggregate_obj_scale_unc(obins, mode), rtol=2e-3), mode         # end to end vs the CPU sampler
    u = sc.hua_score(it['cand'], it['dets'], it['num'], ids, 100, clsW=True, seed=20).cpu().numpy()
    assert np.allclose(u, ohua.aggregate_obj_scale_unc(bins, 'objectSum_scaleMax_classSum', clsW=True), rtol=1e-5)
    # no detections above 0.3 -> score 0 (Lambda_L2.py:615-616)
    z = torch.zeros_like(it['num'])
    assert (sc.hua_score(it['cand'], it['dets'], z, ids, 100).cpu() == 0).all()


def test_nms_many_candidates_path(run):
    """>= 10000 valid (box, class) pairs: mmcv's per-class path; also exercises several sort tranches."""
    sc = run['scoring']
    g = synth.gen(77)
    n, C = 1500, 20
    xy = torch.rand(1, n, 2, generator=g) * 100
    boxes = torch.cat([xy, xy + torch.rand(1, n, 2, generator=g) * 30 + 1], -1)
    scores = torch.rand(1, n, C + 1, generator=g) * 0.5 + 0.06
    scores[..., -1] = 0
    scores[0, ::7, 3] = 0.9            # ties on purpose
    d, lab, keep, num = sc.multiclass_nms_batch(boxes.cuda(), scores.cuda(), 0.05, 0.5, 100)
    od, olab, okeep, _ = odetect.multiclass_nms(boxes[0], scores[0])
    nn = int(num[0])
    assert nn == len(okeep) == 100
    assert np.array_equal(keep[0, :nn].cpu().numpy(), okeep.numpy()) and np.array_equal(lab[0, :nn].cpu().numpy(), olab.numpy())
    # nothing above the threshold
    d, lab, keep, num = sc.multiclass_nms_batch(boxes.cuda(), torch.zeros_like(scores).cuda(), 0.05, 0.5, 100)
    assert int(num[0]) == 0


def test_entropy_all_mode_vs_oracle_and_reference():
    """uncertainty_pool='Entropy_ALL' (every foreground anchor, no top-k / NMS): kernel vs Philox oracle (values) and
    vs the reference's 12 reseeded MC runs (statistics)."""
    from aod_meh_hua_amd import scoring
    from aod_meh_hua_amd.core.anchor import AnchorGenerator
    from aod_meh_hua_amd.core.bbox import DeltaXYWHBBoxCoder
    from oracle import model as om

    class Head:
        last_activation, cls_out_channels, num_anchors = 'relu', 20, 9
        bbox_coder = DeltaXYWHBBoxCoder()
    g = np.load(os.path.join(G, 'scoring_all.npz'))
    cls_p, reg_p, L_p = synth.planted_heads(2, 128, 128)
    mt = synth.metas(2, 128, 128, scale=1.25)
    ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    anchors = ag.grid_anchors([tuple(c.shape[-2:]) for c in cls_p], 'cuda')
    cfg = Cfg(nms_pre=1000, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
    alphas = [om.nhwc_flat(c, 20).softmax(dim=2) for c in cls_p]
    lam = [om.nhwc_flat(l, 1)[..., 0] for l in L_p]
    offs = np.concatenate([[0], np.cumsum([a.shape[1] for a in alphas])[:-1]])
    obins = ohua.compute_scale_unc(alphas, lam, sampler='philox', seed=20, level_offsets=offs)
    umu, usd = g['unc_runs'].mean(0), g['unc_runs'].std(0)
    for mode in ('scaleAvg_classAvg', 'scaleSum_classSum', 'scaleSum_classAvg', 'scaleAvg_classSum'):
        det, unc = scoring.score_batch(Head(), [c.cuda() for c in cls_p], [r.cuda() for r in reg_p], anchors, [m['img_shape'] for m in mt],
                                       [m['scale_factor'] for m in mt], cfg, rescale=True, with_nms=False, isUnc='Epistemic',
                                       uPool='Entropy_ALL', uPool2=mode, isEval=False, L_scores=[l.cuda() for l in L_p], batchIdx=0)
        u = unc.cpu().numpy()
        assert np.allclose(u, ohua.aggregate_scale_unc(obins, mode), rtol=2e-3), (mode, u)
        if mode == 'scaleAvg_classAvg':
            assert (np.abs(u - umu) <= 5 * usd + 0.02 * umu).all(), (u, umu, usd)
    assert det[0][0].shape == (3069, 4) and det[0][1].shape == (3069, 21)


def test_save_max_conf_matches_get_max_conf(run):
    """saveMaxConf third output = getMaxConf (mmdet/utils/functions.py:467-476): max softmax probability over levels / anchors / classes."""
    cand = run['it']['cand']
    cls_p, _, _ = synth.planted_heads(2, 128, 128)
    exp = torch.stack([c.permute(0, 2, 3, 1).reshape(2, -1, 20).softmax(-1).reshape(2, -1).max(-1)[0] for c in cls_p], 1).max(-1)[0]
    assert np.allclose(cand.max_conf().cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('B,H,W,has_bg', [(2, 128, 128, False), (3, 256, 320, False), (16, 512, 512, False), (2, 128, 128, True)])
def test_merged_level_launches_equal_the_per_level_chain(monkeypatch, B, H, W, has_bg):
    """aod_pre_nms_levels (row max of all levels in one launch + one (image, level) workgroup grid for top-k, gather and decode) against the
    13-launch per-level chain on the same head outputs: every output identical (indices, boxes, scores, lambda, level gates)."""
    from aod_meh_hua_amd import scoring
    from aod_meh_hua_amd.core.anchor import AnchorGenerator
    C_ = 21 if has_bg else 20
    cls_p, reg_p, L_p = synth.planted_heads(B, H, W, C=C_, seed=31 + B)
    mt = synth.metas(B, H, W, scale=1.25)
    ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    anchors = ag.grid_anchors([tuple(c.shape[-2:]) for c in cls_p], 'cuda')
    args = ([c.cuda() for c in cls_p], [r.cuda() for r in reg_p], [l.cuda() for l in L_p], anchors, [m['img_shape'] for m in mt],
            [m['scale_factor'] for m in mt], 1000, C_, (0., 0., 0., 0.), (1., 1., 1., 1.))
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('AOD_PRE_NMS_MERGED', mode)
        out[mode] = scoring.pre_nms(*args, has_bg=has_bg)
    torch.cuda.synchronize()
    a, b = out['0'], out['1']
    assert a.level_start == b.level_start
    for name in ('boxes', 'scores', 'lam', 'cand_anchor', 'any_fg'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for x, y in zip(a.topk_idx, b.topk_idx):
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y))
    for x, y in zip(a.rowmax, b.rowmax):
        assert torch.equal(x, y)
    assert torch.equal(a.max_conf(), b.max_conf())


@pytest.mark.parametrize('C,max_num', [(20, 100), (3, 100), (1, 200), (80, 100)])
def test_nms_stress_images_vs_oracle(run, C, max_num):
    """nms_kernel (tranches of 1 024 sorted candidates, greedy scan 64 candidates per round, stop at max_num) against the oracle's statement of
    mmcv's sequential rule (bbox_nms.py:7-93 -> batched_nms) on images that stress it: heavy overlap (several sort tranches, the cut in the
    middle of one), no overlap at all (the first max_num by score), fewer valid entries than max_num, one class keeping more than 64
    candidates, exact score ties, tight clusters; 1 / 3 / 20 / 80 classes (the 80-class image has 112 000 entries, 4x the LDS score cache)."""
    sc = run['scoring']
    g = synth.gen(1234 + C)
    n = 1400
    imgs = []
    for kind in ('dense', 'sparse', 'few', 'oneclass', 'ties', 'clusters'):
        if kind == 'dense':
            xy = torch.rand(n, 2, generator=g) * 40
            wh = torch.rand(n, 2, generator=g) * 30 + 10
        elif kind == 'clusters':
            ctr = torch.rand(12, 2, generator=g) * 400
            xy = ctr[torch.randint(0, 12, (n,), generator=g)] + torch.rand(n, 2, generator=g) * 6
            wh = torch.rand(n, 2, generator=g) * 4 + 20
        else:
            xy = torch.rand(n, 2, generator=g) * 4000
            wh = torch.rand(n, 2, generator=g) * 10 + 2
        boxes = torch.cat([xy, xy + wh], -1)
        scores = torch.rand(n, C + 1, generator=g) * 0.9 + 0.051
        if kind == 'few':
            scores = scores * 0.01
            scores[torch.randint(0, n, (37,), generator=g), torch.randint(0, C, (37,), generator=g)] = 0.5
        if kind == 'oneclass':
            scores[:, 1:] = 0.01
        if kind == 'ties':
            scores = (scores * 8).round() / 8 + 0.06
        scores[:, -1] = 0
        imgs.append((boxes, scores))
    boxes = torch.stack([b for b, _ in imgs]); scores = torch.stack([s for _, s in imgs])
    d, lab, keep, num = sc.multiclass_nms_batch(boxes.cuda(), scores.cuda(), 0.05, 0.5, max_num)
    for i, (b, s) in enumerate(imgs):
        od, olab, okeep, _ = odetect.multiclass_nms(b, s, max_num=max_num)
        nn = int(num[i])
        assert nn == len(okeep), (i, nn, len(okeep))
        assert np.array_equal(keep[i, :nn].cpu().numpy(), okeep.numpy()), i
        assert np.array_equal(lab[i, :nn].cpu().numpy(), olab.numpy()), i
        assert torch.equal(d[i, :nn].cpu(), od), i
        assert bool((keep[i, nn:] == -1).all()) and bool((d[i, nn:] == 0).all())
