"""HIP-graph replay (aod_meh_hua_amd/graphs.py) must do exactly what the eager iteration does: same losses, same parameter updates
(up to the fp32 atomics of wgrad / column sums), same scores, and it must follow a changing learning rate and new input data."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(params=['bf16', 'bf16x3'])
def arithmetic(request):
    """The replayed step must equal the eager one.  In the fast mode the fp32 atomics of the column sums never reach an operand (8-bit
    roundings absorb them), so the two agree for as long as one cares to look.  In the reference-precision mode a 1e-7 difference in a bias DOES
    reach the 16-bit operands, and these tests' deliberately large steps amplify it: there the comparison runs in the
    deterministic mode (ordered column sums), where eager and replayed iterations are the same bits."""
    from aod_meh_hua_amd import functional as AF
    AF.set_precision(request.param)
    if request.param == 'bf16x3':
        AF.set_deterministic(True)
    yield request.param
    AF.set_deterministic(False)
    AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))


def _build(lr=2e-4):
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    from aod_meh_hua_amd.optim import FusedSGD
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(cls_bias=-2.0), strict=True)
    model = model.cuda().train()
    head = model.bbox_head
    meh = set(id(p) for n in ('retina_L', 'L_convs') for p in getattr(head, n).parameters())
    main = [p for p in model.parameters() if p.requires_grad and id(p) not in meh]
    opt = FusedSGD(main, lr=lr, momentum=0.9, weight_decay=1e-4)      # small steps: the comparison must not be chaotic
    opt_L = FusedSGD([p for p in model.parameters() if id(p) in meh], lr=lr, momentum=0.9, weight_decay=1e-4)
    return model, opt, opt_L


def _batch(seed, B=2, H=128):
    gtb, gtl = synth.random_gts(B, H, H, seed=seed, gmin=1, gmax=3)
    return dict(img=synth.images(B, H, H, seed=seed).cuda(), img_metas=synth.metas(B, H, H), gt_bboxes=gtb, gt_labels=gtl)


def _eager_iter(model, opt, opt_L, data):
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad()
    out['loss'].backward()
    opt.step()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad()
    lossL['loss'].backward()
    opt_L.step()
    return float(out['loss'].detach()), float(lossL['loss'].detach())


def test_graphed_train_step_equals_eager(arithmetic):
    from aod_meh_hua_amd.graphs import GraphedTrainStep
    batches = [_batch(31), _batch(32), _batch(33)]
    lrs = [2e-4, 2e-4, 5e-5]
    # eager reference
    model, opt, opt_L = _build()
    ref_losses = []
    for d, lr in zip(batches, lrs):
        opt.param_groups[0]['lr'] = lr
        ref_losses.append(_eager_iter(model, opt, opt_L, d))
    ref = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    # graphed: the capture happens inside the first call; its warm-up iterations are real updates that GraphedTrainStep must undo
    # (parameters, momentum, BN buffers snapshotted / restored), so the first call applies batch 0 exactly once
    model2, opt2, opt_L2 = _build()
    sd0 = {k: v.detach().clone() for k, v in model2.state_dict().items()}
    gs = GraphedTrainStep(model2, opt2, opt_L2, warmup=2, Labeled=True, Pseudo=False)
    got = []
    for d, lr in zip(batches, lrs):
        opt2.param_groups[0]['lr'] = lr
        o = gs(d)
        got.append((float(o['loss']), float(o['log_vars']['loss_L'])))
    torch.cuda.synchronize()
    assert np.allclose(np.array(got), np.array(ref_losses), rtol=2e-3), (got, ref_losses)
    new = {k: v.detach().float().cpu() for k, v in model2.state_dict().items()}
    worst = max(float((new[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)) for k in ref if ref[k].is_floating_point())
    assert worst < 5e-3, worst
    moved = float((new['bbox_head.retina_cls.weight'] - sd0['bbox_head.retina_cls.weight'].float().cpu()).abs().max())
    assert moved > 0
    # eager code after replays sees the updated parameters (version bump -> packed weights rebuilt)
    model2.train()
    l_eager = float(model2.train_step(batches[0], Labeled=True, Pseudo=False)[0]['loss'].detach())
    model.train()
    l_ref = float(model.train_step(batches[0], Labeled=True, Pseudo=False)[0]['loss'].detach())
    assert np.allclose(l_eager, l_ref, rtol=2e-3), (l_eager, l_ref)


def test_graphed_score_equals_eager_and_follows_inputs():
    from aod_meh_hua_amd.graphs import GraphedScore
    model, _, _ = _build()
    model.eval()
    kw = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
              showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
    gsc = GraphedScore(model, **kw)
    metas = synth.metas(2, 128, 128)
    for seed in (41, 42, 43):
        img = synth.images(2, 128, 128, seed=seed).cuda()
        ids = torch.tensor([seed * 2, seed * 2 + 1], device='cuda')
        with torch.no_grad():
            _, unc_e = model(img=[img], img_metas=[metas], return_loss=False, image_ids=ids, **kw)
        _, unc_g = gsc(img, metas, ids)
        assert torch.equal(torch.as_tensor(unc_e).float().cpu(), unc_g.float().cpu()), (seed, unc_e, unc_g)
    assert gsc.maybe(synth.images(1, 128, 128).cuda(), synth.metas(1, 128, 128), torch.zeros(1, dtype=torch.int64, device='cuda')) is None


def test_capture_applies_the_batch_once_and_eager_iterations_may_interleave(arithmetic):
    """(a) maybe() returns None for a first-seen shape, captures on its second consecutive appearance and from then on replays
    it from the cache, also after other shapes ran eagerly in between; (b) every path applies exactly one update per batch: the sequence
    graph / eager / graph equals three eager iterations; (c) an eager iteration between replays (zero_grad(set_to_none) rebinding .grad)
    does not disturb later replays."""
    from aod_meh_hua_amd.graphs import GraphedTrainStep
    seq = [_batch(51), _batch(52), _batch(53, B=1), _batch(54), _batch(55, B=1), _batch(56)]
    model, opt, opt_L = _build()
    ref = [_eager_iter(model, opt, opt_L, d) for d in seq]
    ref_sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    model2, opt2, opt_L2 = _build()
    gs = GraphedTrainStep(model2, opt2, opt_L2, warmup=2, Labeled=True, Pseudo=False)
    got, how = [], []
    for d in seq:
        o = gs.maybe(d)
        if o is None:
            how.append('eager')
            got.append(_eager_iter(model2, opt2, opt_L2, d))
        else:
            how.append('graph')
            got.append((float(o['loss']), float(o['log_vars']['loss_L'])))
    torch.cuda.synchronize()
    assert how == ['eager', 'graph', 'eager', 'graph', 'eager', 'graph'], how       # B=2 shape: captured once, replayed from the cache
    assert len(gs.cache) == 1
    assert np.allclose(np.array(got), np.array(ref), rtol=3e-3), (got, ref)
    new = {k: v.detach().float().cpu() for k, v in model2.state_dict().items()}
    worst = max(float((new[k] - ref_sd[k]).abs().max() / (ref_sd[k].abs().max() + 1e-12)) for k in ref_sd if ref_sd[k].is_floating_point())
    assert worst < 5e-3, worst


def _ragged_batch(seed, shapes, H=128, W=160):
    """two images zero-padded into one [2, 3, H, W] tensor with their own img / pad shapes and scale factors (what mmcv-style collate
    hands over for keep-ratio data)"""
    B = len(shapes)
    img = torch.zeros(B, 3, H, W)
    metas, gtb, gtl = [], [], []
    for b, (h, w, ph, pw, sf) in enumerate(shapes):
        img[b, :, :h, :w] = synth.images(1, h, w, seed=seed + b)[0]
        metas.append(dict(img_shape=(h, w, 3), pad_shape=(ph, pw, 3), ori_shape=(int(h / sf), int(w / sf), 3),
                          scale_factor=np.array([sf] * 4, np.float32), flip=False, flip_direction=None))
        bb, ll = synth.random_gts(1, h, w, seed=seed + 10 + b, gmin=1, gmax=3)
        gtb.append(bb[0]), gtl.append(ll[0])
    return dict(img=img.cuda(), img_metas=metas, gt_bboxes=gtb, gt_labels=gtl)


def test_one_graph_serves_batches_that_differ_only_in_their_per_image_shapes(arithmetic):
    """keep-ratio VOC batches of one padded tensor shape differ in their per-image pad shapes (valid-anchor flags), image
    sizes and scale factors (box clipping / rescaling in the scoring pass).  Those are STATIC device inputs of the captured graphs
    (L_AnchorHead.get_targets_batch, scoring.static_meta): the second batch REPLAYS the graph captured on the first and must equal eager."""
    from aod_meh_hua_amd.graphs import GraphedScore, GraphedTrainStep
    a = _ragged_batch(71, [(128, 160, 128, 160, 1.0), (128, 160, 128, 160, 1.0)])
    b = _ragged_batch(75, [(120, 150, 128, 160, 1.25), (90, 100, 96, 128, 0.8)])
    model, opt, opt_L = _build()
    ref = [_eager_iter(model, opt, opt_L, d) for d in (a, b)]
    ref_sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    model2, opt2, opt_L2 = _build()
    gs = GraphedTrainStep(model2, opt2, opt_L2, warmup=1, Labeled=True, Pseudo=False)
    got = []
    for d in (a, b):
        o = gs(d)
        got.append((float(o['loss']), float(o['log_vars']['loss_L'])))
    torch.cuda.synchronize()
    assert len(gs.cache) == 1, 'the second batch must replay the first batch\'s graph'
    assert np.allclose(np.array(got), np.array(ref), rtol=3e-3), (got, ref)
    new = {k: v.detach().float().cpu() for k, v in model2.state_dict().items()}
    worst = max(float((new[k] - ref_sd[k]).abs().max() / (ref_sd[k].abs().max() + 1e-12)) for k in ref_sd if ref_sd[k].is_floating_point())
    assert worst < 5e-3, worst
    # scoring: image sizes / scale factors are static inputs too
    model.eval()
    kw = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
              showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
    with torch.no_grad():            # trained-like head so that the scores are not all zero
        model.bbox_head.retina_cls.weight.mul_(60.0), model.bbox_head.retina_cls.bias.mul_(60.0)
    gsc = GraphedScore(model, **kw)
    for d, base in ((a, 0), (b, 2), (a, 4)):
        ids = torch.tensor([base, base + 1], device='cuda')
        with torch.no_grad():
            _, unc_e = model(img=[d['img']], img_metas=[d['img_metas']], return_loss=False, image_ids=ids, **kw)
        _, unc_g = gsc(d['img'], d['img_metas'], ids)
        assert torch.equal(torch.as_tensor(unc_e).float().cpu(), unc_g.float().cpu()), (base, unc_e, unc_g)
    assert len(gsc.cache) == 1 and float(unc_g.abs().sum()) > 0


def test_multi_scale_shape_sequence_through_the_graph_cache(arithmetic):
    """BASELINE configs[4] trains multi-scale (short side drawn per batch from {640 ... 800},
    configs/retinanet/retinanet_r50_caffe_fpn_mstrain_1x_coco.py:15-19 of the reference): the padded batch shape changes from iteration to
    iteration and comes back.  Scaled-down shapes of the same aspect ratios through GraphedTrainStep.maybe(): a shape is run eagerly when first
    seen, captured when it repeats, replayed from the cache when it returns after other shapes -- and the sequence equals the all-eager run."""
    from aod_meh_hua_amd.graphs import GraphedTrainStep
    shapes = {'a': (160, 272), 'b': (176, 288), 'c': (200, 336)}
    order = ['a', 'a', 'b', 'b', 'c', 'c', 'a', 'b', 'c']

    def batch(seed, hw):
        H, W = hw
        gtb, gtl = synth.random_gts(2, H, W, seed=seed, gmin=1, gmax=3)
        return dict(img=synth.images(2, H, W, seed=seed).cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=gtb, gt_labels=gtl)
    seq = [batch(80 + i, shapes[k]) for i, k in enumerate(order)]
    model, opt, opt_L = _build(lr=1e-5)          # (nine iterations: at the other tests' step size the seeded network's loss diverges)
    ref = [_eager_iter(model, opt, opt_L, d) for d in seq]
    model2, opt2, opt_L2 = _build(lr=1e-5)
    gs = GraphedTrainStep(model2, opt2, opt_L2, warmup=1, Labeled=True, Pseudo=False)
    got, how = [], []
    for d in seq:
        o = gs.maybe(d)
        if o is None:
            how.append('eager')
            got.append(_eager_iter(model2, opt2, opt_L2, d))
        else:
            how.append('graph')
            got.append((float(o['loss']), float(o['log_vars']['loss_L'])))
    torch.cuda.synchronize()
    assert how == ['eager', 'graph', 'eager', 'graph', 'eager', 'graph', 'graph', 'graph', 'graph'], how
    assert len(gs.cache) == 3
    assert np.allclose(np.array(got), np.array(ref), rtol=3e-3), (got, ref)
