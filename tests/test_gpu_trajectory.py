"""GPU: a multi-iteration TRAINING TRAJECTORY.  30 consecutive `run_iter`s (Epoch_Based_Runner_Lambda.py:20-38: main
forward / backward / SGD, then the MEH step) of the HIP model in both precision modes beside the fp32 CPU oracle, from the same seeded
weights on the same four fixed batches (B = 2, 128^2, momentum 0.9, weight decay 1e-4; lr 1e-5: the seeded N(0, 0.02) weights start with a
gradient norm of ~750, at the config's 1e-3 the ORACLE diverges to NaN within four steps): the loss curves stay inside a stated band
at EVERY step and the parameters end up where the oracle's do."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS, B, S, LR = 30, 2, 128, 1e-5


def _batches():
    out = []
    for k in range(4):
        gtb, gtl = synth.random_gts(B, S, S, seed=100 + k, gmin=1, gmax=3)
        out.append((synth.images(B, S, S, seed=50 + k), gtb, gtl))
    return out


def _oracle_run(batches):
    sd = omodel.seeded_state_dict()
    keys = [k for k, v in sd.items() if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.'))]
    for k in keys:
        sd[k].requires_grad_(True)
    meh = [k for k in keys if 'retina_L' in k or 'L_convs' in k]
    main = [k for k in keys if k not in meh]
    bufs, bufs_L, losses, losses_L = {}, {}, [], []
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    for it in range(ITERS):
        img, gtb, gtl = batches[it % len(batches)]
        o = omodel.train_step(sd, img, gtb, gtl)
        for k in keys:
            sd[k].grad = None
        o['loss'].backward()
        with torch.no_grad():
            omodel.sgd_step({k: sd[k] for k in main}, {k: sd[k].grad for k in main}, bufs, lr=LR)
        oL = omodel.train_step_L(sd, o['feats'], o['loss_noR'], o['targets'])
        for k in keys:
            sd[k].grad = None
        oL['loss'].backward()
        with torch.no_grad():
            omodel.sgd_step({k: sd[k] for k in meh}, {k: sd[k].grad for k in meh}, bufs_L, lr=LR)
        losses.append(float(o['loss']))
        losses_L.append(float(oL['loss']))
    return np.array(losses), np.array(losses_L), {k: v.detach().clone() for k, v in sd.items()}


def _hip_run(batches, prec):
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    from aod_meh_hua_amd.optim import FusedSGD
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(), strict=True)
    model = model.cuda().train()
    mehp = [p for n, p in model.named_parameters() if ('retina_L' in n or 'L_convs' in n)]
    ids = {id(p) for p in mehp}
    mainp = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    opt, opt_L = FusedSGD(mainp, lr=LR, momentum=0.9, weight_decay=1e-4), FusedSGD(mehp, lr=LR, momentum=0.9, weight_decay=1e-4)
    losses, losses_L = [], []
    AF.set_precision(prec)
    try:
        for it in range(ITERS):
            img, gtb, gtl = batches[it % len(batches)]
            data = dict(img=img.cuda(), img_metas=synth.metas(B, S, S), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
            out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
            opt.zero_grad()
            out['loss'].backward()
            opt.step()
            lossL = model.train_step_L(prev, head_out, feat_out)
            opt_L.zero_grad()
            lossL['loss'].backward()
            opt_L.step()
            losses.append(float(out['loss'].detach()))
            losses_L.append(float(lossL['loss'].detach()))
        torch.cuda.synchronize()
    finally:
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
    return np.array(losses), np.array(losses_L), {k: v.detach().float().cpu() for k, v in model.state_dict().items()}


def test_thirty_iterations_beside_the_fp32_oracle():
    batches = _batches()
    lo, lLo, sdo = _oracle_run(batches)
    sd0 = omodel.seeded_state_dict()
    names = ['backbone.layer2.0.conv1.weight', 'backbone.layer4.2.conv3.weight', 'neck.fpn_convs.0.conv.weight', 'bbox_head.cls_convs.3.conv.weight',
             'bbox_head.retina_cls.weight', 'bbox_head.retina_cls.bias', 'bbox_head.retina_L.weight']
    assert lo[-4:].mean() < lo[:4].mean(), 'the oracle trajectory does not train'          # (the run is a real optimisation, not a fixed point)
    res = {}
    for prec in ('bf16x3', 'bf16'):
        l, lL, sd = _hip_run(batches, prec)
        dl, dlL = np.abs(l - lo) / np.abs(lo), np.abs(lL - lLo) / np.maximum(np.abs(lLo), 1e-12)
        # drift: distance to the oracle's final parameters relative to the distance the oracle travelled from the initial ones
        drift = {k: float((sd[k] - sdo[k]).norm() / ((sdo[k] - sd0[k]).norm() + 1e-30)) for k in names}
        res[prec] = (dl, dlL, drift)
        print(f'\n{prec}: loss deviation max {dl.max():.2e} (step {int(dl.argmax())}), last {dl[-1]:.2e}; MEH loss deviation max {dlL.max():.2e}; '
              f'parameter drift {({k.split(".", 1)[1]: round(v, 5) for k, v in drift.items()})}')
    dl, dlL, drift = res['bf16x3']
    # reference precision: on the oracle's curve at every one of the 30 steps; the parameters moved where the oracle's moved
    # (measured: 1.9e-3 at step 27 -- per-step differences of ~1e-5 grow along a 30-step trajectory; the fast mode reaches 1.7e-2)
    assert dl.max() < 5e-3 and dlL.max() < 5e-3, (dl.max(), dlL.max())
    assert max(drift.values()) < 2e-2, drift
    dl16, dlL16, drift16 = res['bf16']
    # fast mode: inside a 5 % band at every step (measured: 1.7e-2 at worst, 0.4 % in the first ten steps and ~1.3 % in the last five --
    # the deviation grows along the trajectory, without running away)
    assert dl16.max() < 5e-2, dl16
    assert dl16[-5:].max() < 10 * max(dl16[:10].max(), 1e-3), dl16
    assert max(drift16.values()) < 0.5, drift16
    assert dl.max() < dl16.max() / 5
