"""GPU, end to end through the REAL conv stack: what each precision mode does to detections, image scores and the
active-learning selection, asserted against the fp32 CPU oracle on the full seeded model with a trained-like (scaled) classification head.

  * `bf16x3` (reference precision, the bench headline): every oracle detection is found (box 0.5 px, score 1e-3), image scores agree to 1 %,
    and `update_X_L` selects the oracle's images up to swaps of near-tied images at the selection boundary.
  * `bf16` (fast mode): bounded, and visibly looser -- operand rounding moves logits by ~1e-2 relative, which moves anchors across the 0.3
    foreground gate, candidates across the 0.5 IoU gate of `GetObjectIdx` (Lambda_L2.py:343-361) and near-tied candidates in and out of the
    100 kept detections: image scores move by 1-20 %, the selection of 16 of 56 pool images differs from the oracle's in three images (rank
    correlation 0.93).  The asserted bounds carry a ~2x margin over what was measured; they are the price of the fast mode and the reason the
    bench headline is not measured in it.

Reference path: tools/train_RetinaNet.py:221-251 -> mmdet/apis/test.py:65-135 -> Lambda_L2.py:254-384,489-619 -> mmdet/utils/active_datasets.py:102-135."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
          showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)


def _calibrated(img):
    """seeded model whose retina_cls is scaled until ~1 % of the anchors are foreground (> 0.3) on `img` (calibrated on the ORACLE's logits)"""
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    sd = omodel.seeded_state_dict()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        cls, _ = omodel.head_forward(sd, omodel.fpn(sd, omodel.backbone(sd, img)))
        x = torch.cat([omodel.nhwc_flat(c, 20) for c in cls], 1)
        lo, hi = 1.0, 1e5
        for _ in range(40):
            mid = (lo * hi) ** 0.5
            lo, hi = (mid, hi) if float((torch.softmax(x * mid, -1).amax(-1) > 0.3).float().mean()) < 0.01 else (lo, mid)
        k = float(np.float32(hi))
        for key in ('bbox_head.retina_cls.weight', 'bbox_head.retina_cls.bias'):
            sd[key] = sd[key] * k
    model.load_state_dict(sd, strict=True)
    return model.cuda().eval(), sd


def _hip_scores(model, img, ids, prec):
    from aod_meh_hua_amd import functional as AF
    AF.set_precision(prec)
    try:
        with torch.no_grad():
            B, _, H, W = img.shape
            dets, unc = model(img=[img.cuda()], img_metas=[synth.metas(B, H, W)], return_loss=False, image_ids=ids.cuda(), **KW)
        torch.cuda.synchronize()
    finally:
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
    return torch.as_tensor(unc).float().cpu().numpy().astype(np.float64), dets


def _matched(oracle_dets, hip_dets, box_tol, score_tol):
    """fraction of the oracle's detections for which the HIP pass has a detection of the same class with the same box and score"""
    hit = tot = 0
    for (od, olab, _), (gd, glab) in zip(oracle_dets, hip_dets):
        gd, glab = torch.as_tensor(gd).float().cpu(), torch.as_tensor(glab).cpu().long()
        for k in range(od.shape[0]):
            same = (glab == int(olab[k])) & ((gd[:, :4] - od[k, :4]).abs().amax(1) < box_tol) & ((gd[:, 4] - od[k, 4]).abs() < score_tol)
            hit += int(same.any())
            tot += 1
    return hit / max(tot, 1), tot


@pytest.mark.parametrize('B,S', [(4, 128), (2, 512)])
def test_detections_and_image_scores_of_both_modes_against_the_oracle(B, S):
    img = synth.images(B, S, S, seed=41)
    model, sd = _calibrated(img)
    mt = synth.metas(B, S, S)
    ids = torch.arange(B)
    o = omodel.score_images(sd, img, [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt], sampler='philox', seed=20, image_ids=ids.numpy())
    ref = np.array(o['unc'], np.float64)
    assert (ref > 0).all(), ref
    u3, d3 = _hip_scores(model, img, ids, 'bf16x3')
    u16, d16 = _hip_scores(model, img, ids, 'bf16')
    m3, n = _matched(o['dets'], d3, 0.5, 1e-3)
    m16, _ = _matched(o['dets'], d16, 0.5, 2e-2)
    print('bf16 matched at (px, score):', {t: round(_matched(o['dets'], d16, *t)[0], 3) for t in ((0.5, 2e-2), (1.0, 5e-2), (2.0, 0.1), (4.0, 0.25), (1e9, 1e9))},
          'bf16x3 at (0.5, 1e-4):', round(_matched(o['dets'], d3, 0.5, 1e-4)[0], 4), 'detections per image', [int(torch.as_tensor(d[0]).shape[0]) for d in d16])
    dev3, dev16 = np.abs(u3 - ref) / ref, np.abs(u16 - ref) / ref
    print(f'\n{B}x{S}^2: {n} oracle detections; matched bf16x3 {m3:.4f} (0.5 px, 1e-3)  bf16 {m16:.4f} (0.5 px, 2e-2); '
          f'image-score deviation bf16x3 {dev3}  bf16 {dev16}')
    # reference precision: the same detections (the last few of the 100 kept per image may trade places with the first few not kept) and
    # the same scores (a (candidate, object) pair within rounding distance of the 0.5 IoU gate may fall on the other side: ~0.3 % per pair)
    assert m3 >= 1 - 3.0 * B / n, (m3, n)
    assert dev3.max() < 1e-2 and np.median(dev3) < 1e-3, dev3
    # fast mode: it does NOT meet the (0.5 px, 2e-2) criterion -- measured: 54 % of the oracle's detections at 4 x 128^2 and 8.5 % at 2 x 512^2
    # (the 100 kept per image are cut out of thousands of near-tied candidates; 97 % match at (4 px, 0.25) at 128^2) -- and moves image scores
    # by 1-12 %.  What holds, with margin:
    loose, _ = _matched(o['dets'], d16, 4.0, 0.25)
    assert loose >= 0.5, loose
    assert dev16.max() < 0.25, dev16
    assert dev3.max() < dev16.max() / 5


def test_pool_selection_through_the_conv_stack():
    """64 seeded 128^2 images scored batch by batch through backbone, neck, heads, top-k, NMS and HUA; update_X_L (8 labeled, 16 to select)
    on the HIP scores of either mode against the selection from the oracle's scores"""
    from aod_meh_hua_amd.utils.active_datasets import update_X_L
    N, S, bs = 64, 128, 8
    imgs = synth.images(N, S, S, seed=77)
    model, sd = _calibrated(imgs[:8])
    mt = synth.metas(bs, S, S)
    ref, u3, u16 = [], [], []
    for s in range(0, N, bs):
        ids = torch.arange(s, s + bs)
        o = omodel.score_images(sd, imgs[s:s + bs], [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt], sampler='philox', seed=20,
                                image_ids=ids.numpy())
        ref.append(np.array(o['unc'], np.float64))
        u3.append(_hip_scores(model, imgs[s:s + bs], ids, 'bf16x3')[0])
        u16.append(_hip_scores(model, imgs[s:s + bs], ids, 'bf16')[0])
    ref, u3, u16 = np.concatenate(ref), np.concatenate(u3), np.concatenate(u16)

    def select(u):
        np.random.seed(20)
        return update_X_L(u, np.arange(N), np.arange(8), 16, zeroRate=0.15)[0]
    s_ref, s3, s16 = select(ref), select(u3), select(u16)
    pool = np.setdiff1d(np.arange(N), np.arange(8))
    order = pool[np.argsort(ref[pool])]
    n_top = 16 - int(16 * 0.15)          # update_X_L: the top (X_S_size - int(X_S_size * zeroRate)) scores + up to that many zero-score images
    boundary = ref[order[-n_top]]
    d3, d16 = sorted(set(s_ref) ^ set(s3)), sorted(set(s_ref) ^ set(s16))
    rank = lambda u: np.argsort(np.argsort(u[pool]))
    rho3 = float(np.corrcoef(rank(ref), rank(u3))[0, 1])
    rho16 = float(np.corrcoef(rank(ref), rank(u16))[0, 1])
    print(f'\npool of {N}: score deviation bf16x3 max {np.abs(u3 - ref).max() / ref.max():.2e}  bf16 max {np.abs(u16 - ref).max() / ref.max():.2e}; '
          f'selection differences vs the oracle: bf16x3 {d3}  bf16 {d16}; rank correlation bf16x3 {rho3:.4f}  bf16 {rho16:.4f}')
    # reference precision: the oracle's selection, up to swaps of images whose oracle scores lie within 2 % of the boundary score
    assert len(d3) <= 2 and all(abs(ref[i] - boundary) < 0.02 * boundary for i in d3), (d3, ref[d3], boundary)
    assert rho3 > 0.999 and np.abs(u3 - ref).max() / ref.max() < 1e-2
    # fast mode: the ranking survives in the large (rank correlation), the selected set may differ in several images
    # (measured: rank correlation 0.928, three of the sixteen selected images differ, scores off by up to 20 % of the largest)
    assert rho16 > 0.8 and len(d16) <= 12, (rho16, d16)
    assert np.abs(u16 - ref).max() / ref.max() < 0.4
