"""GPU: evaluation path end to end -- single_gpu_test(isEval=True) -> bbox2result -> dataset.evaluate (fork metric) via EvalHook."""
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_hook_runs_the_fork_metric():
    from aod_meh_hua_amd.datasets import build_dataloader, build_dataset
    from aod_meh_hua_amd.mmcv_lite import Config, EvalHook, LogBuffer, MMDataParallel
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(cls_bias=1.0), strict=True)      # plenty of (random) detections above score_thr
    model = MMDataParallel(model.cuda())
    ds = build_dataset(dict(type='SyntheticVOCDataset', num_images=6, size=(128, 128)), dict(test_mode=True))
    dl = build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=0, dist=False, shuffle=False)

    class R:
        epoch, logger = 4, None
        log_buffer = LogBuffer()
    R.model = model
    hook = EvalHook(dl, interval=5, metric='mAP', show=False, isUnc=False, out_dir=None)
    res = hook.after_train_epoch(R)
    assert res is not None and 0.0 <= res['mAP'] <= 1.0 and 'AP50' in res
    assert R.log_buffer.output['mAP'] == res['mAP'] and R.log_buffer.output['eval_iter_num'] == 3
    R.epoch = 5
    assert hook.after_train_epoch(R) is None          # (epoch + 1) % interval != 0
    # results format: per image a list of num_classes (k, 5) arrays
    from aod_meh_hua_amd.apis.test import single_gpu_test
    out = single_gpu_test(model, dl, isUnc=False)
    assert len(out) == 6 and len(out[0]) == 20 and all(a.shape[1] == 5 for a in out[0])
    assert sum(a.shape[0] for img in out for a in img) > 0
