"""GPU: the real VOC data path feeds the HIP hot path -- padded variable-shape batches through train_step / backward, the sharded pool
scoring loop and evaluation."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel
from tests.test_voc_data import TEST, TRAIN, voc  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(params=['bf16', 'bf16x3'])
def precision(request):
    from aod_meh_hua_amd import functional as AF
    AF.set_precision(request.param)
    yield request.param
    AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))


def test_voc_batches_through_train_score_eval(voc, precision):  # noqa: F811
    from aod_meh_hua_amd.apis.test import single_gpu_test, single_gpu_uncertainty
    from aod_meh_hua_amd.datasets import build_dataloader, build_dataset
    from aod_meh_hua_amd.mmcv_lite import Config, MMDataParallel
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(cls_bias=1.0), strict=True)
    model = MMDataParallel(model.cuda())
    ann = voc + 'ImageSets/Main/trainval.txt'
    train = build_dataset(dict(type='VOCDataset', ann_file=[ann, ann], img_prefix=[voc, voc], pipeline=TRAIN))
    np.random.seed(0)
    dl = build_dataloader(train, samples_per_gpu=2, workers_per_gpu=0, dist=False, shuffle=True, seed=0)
    model.train()
    shapes = set()
    for i, batch in enumerate(dl):
        out, head_out, feat_out, prev = model.train_step(batch, Labeled=True, Pseudo=False)
        model.zero_grad()
        out['loss'].backward()
        assert torch.isfinite(out['loss']).item()
        shapes.add(tuple(batch['img'].data[0].shape))
        if i == 2:
            break
    assert len(shapes) >= 2                                  # landscape and portrait groups: different padded shapes
    g = model.module.bbox_head.retina_cls.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # pool scoring on the (train-pipeline) test split, as the reference configs do (data.test uses train_pipeline)
    pool = build_dataset(dict(type='VOCDataset', ann_file=ann, img_prefix=voc, pipeline=TRAIN), dict(test_mode=False))
    pdl = build_dataloader(pool, samples_per_gpu=1, workers_per_gpu=0, dist=False, shuffle=False)
    unc = single_gpu_uncertainty(model, pdl, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', showNMS=False,
                                 saveUnc=False, saveMaxConf=False, clsW=False)
    assert unc.shape == (len(pool),) and torch.isfinite(unc).all()
    # evaluation split with the test pipeline
    val = build_dataset(dict(type='VOCDataset', ann_file=ann, img_prefix=voc, pipeline=TEST), dict(test_mode=True))
    vdl = build_dataloader(val, samples_per_gpu=1, workers_per_gpu=0, dist=False, shuffle=False)
    res = single_gpu_test(model, vdl, isUnc=False)
    ev = val.evaluate(res, metric='mAP', logger='silent')
    assert len(res) == len(val) and 0.0 <= ev['mAP'] <= 1.0


def test_pool_loader_with_worker_prefetch_scores_like_the_synchronous_loop(voc):  # noqa: F811
    """apis/test.py _shard_batches: worker processes + pinned prefetch (the reference's pool loader has cfg.data.workers_per_gpu workers,
    tools/train_RetinaNet.py:224-225) must hand the model the same batches in the same order as the synchronous loop."""
    from aod_meh_hua_amd.apis.test import single_gpu_uncertainty
    from aod_meh_hua_amd.datasets import build_dataloader, build_dataset
    from aod_meh_hua_amd.mmcv_lite import Config, MMDataParallel
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(cls_bias=1.0), strict=True)
    model = MMDataParallel(model.cuda())
    ann = voc + 'ImageSets/Main/trainval.txt'
    noflip = [dict(t, flip_ratio=0.0) if t['type'] == 'RandomFlip' else t for t in TRAIN]
    pool = build_dataset(dict(type='VOCDataset', ann_file=[ann] * 4, img_prefix=[voc] * 4, pipeline=noflip), dict(test_mode=False))
    kw = dict(isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False)
    outs = []
    for workers in (0, 2):
        pdl = build_dataloader(pool, samples_per_gpu=2, workers_per_gpu=workers, dist=False, shuffle=False)
        outs.append(single_gpu_uncertainty(model, pdl, **kw).cpu())
    assert outs[0].shape == (len(pool),) and torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
