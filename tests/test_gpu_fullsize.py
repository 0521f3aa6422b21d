"""GPU, BASELINE.json's full size (16 images of 512 x 512, 49 104 anchors each): size-independent properties -- the forward loss is
reproducible bit for bit, HUA scores do not depend on how the pool is cut into batches, targets obey the assigner's invariants."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
          showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False)


@pytest.fixture(scope='module')
def model():
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    m = build_detector(cfg.model)
    m.load_state_dict(omodel.seeded_state_dict(cls_bias=-1.0), strict=True)
    return m.cuda()


def test_full_size_train_step_properties(model):
    B, H = 16, 512
    gtb, gtl = synth.random_gts(B, H, H, seed=5, gmin=1, gmax=5)
    data = dict(img=synth.images(B, H, H, seed=6).cuda(), img_metas=synth.metas(B, H, H), gt_bboxes=[b.cuda() for b in gtb],
                gt_labels=[l.cuda() for l in gtl])
    model.train()
    out1, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    out2, *_ = model.train_step(data, Labeled=True, Pseudo=False)
    assert torch.equal(out1['loss'].detach(), out2['loss'].detach())                       # forward is deterministic
    labels = torch.cat([l.reshape(B, -1) for l in head_out[4]], 1)
    bw = torch.cat([w.reshape(B, -1, 4) for w in head_out[7]], 1)
    assert labels.shape == (B, 49104) and int(labels.min()) >= 0 and int(labels.max()) == 20
    pos = labels < 20
    assert torch.equal(pos, bw[..., 0] > 0) and int(head_out[8]) == int(pos.sum(1).clamp(min=1).sum())
    for b in range(B):                                                                     # positives only carry labels of that image's gts
        assert set(labels[b][pos[b]].unique().tolist()) <= set(gtl[b].tolist())
        assert int(pos[b].sum()) >= len(gtb[b])                                            # every gt keeps at least its best anchor (min_pos_iou = 0)
    model.zero_grad()
    out1['loss'].backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_full_size_scores_do_not_depend_on_batching(model):
    B, H = 16, 512
    img = synth.images(B, H, H, seed=9).cuda()
    metas = synth.metas(B, H, H)
    ids = torch.arange(100, 100 + B, device='cuda')
    model.eval()
    with torch.no_grad():
        _, u16 = model(img=[img], img_metas=[metas], return_loss=False, image_ids=ids, **KW)
        _, ua = model(img=[img[:8]], img_metas=[metas[:8]], return_loss=False, image_ids=ids[:8], **KW)
        _, ub = model(img=[img[8:]], img_metas=[metas[8:]], return_loss=False, image_ids=ids[8:], **KW)
        _, u16b = model(img=[img], img_metas=[metas], return_loss=False, image_ids=ids, **KW)
    u16, u16b = torch.as_tensor(u16).float().cpu(), torch.as_tensor(u16b).float().cpu()
    cut = torch.cat([torch.as_tensor(ua).float().cpu(), torch.as_tensor(ub).float().cpu()])
    assert torch.equal(u16, u16b) and torch.isfinite(u16).all()
    assert torch.equal(u16, cut), (u16, cut)
