"""train_detector_SSL (mmdet/apis/train_Lambda.py:37-109): loaders, MMDataParallel wrap, the main SGD optimizer with
the MEH parameters removed + `optimizer_L` for the MEH parameters, training hooks (OptimizerHook dropped), run_SSL."""
from operator import itemgetter

import torch

from ..datasets import build_dataloader
from ..mmcv_lite import MMDataParallel, build_runner
from ..optim import FusedSGD, build_optimizer
from ..parallel import broadcast_model
from ..utils import get_root_logger
from ..utils import Epoch_Based_Runner_Lambda  # noqa: F401  (registers MyEpochBasedRunnerLambda)


def RemoveParamFromOptim(optimizer, model, param_name):
    """train_Lambda.py:97-109."""
    targetIDs = [id(p) for n, p in model.named_parameters() if param_name in n]
    params = optimizer.param_groups[0]['params']
    optimizer.param_groups[0]['params'] = [p for p in params if id(p) not in targetIDs]


def train_detector_SSL(model, dataset, cfg, distributed=False, validate=False, timestamp=None, meta=None):
    logger = get_root_logger(cfg.log_level)
    dataset = dataset if isinstance(dataset, (list, tuple)) else [dataset]
    data_loaders = [build_dataloader(ds, cfg.data.samples_per_gpu, cfg.data.workers_per_gpu, len(cfg.gpu_ids), dist=distributed,
                                     seed=cfg.seed) for ds in dataset]
    dev = torch.device('cuda', torch.cuda.current_device())
    model = MMDataParallel(model.to(dev), device_ids=cfg.gpu_ids)
    broadcast_model(model.module)
    optimizer = build_optimizer(model, cfg.optimizer)
    head = model.module.bbox_head
    meh_names = getattr(head, 'L_names', ['retina_L', 'L_convs'])      # train_Lambda.py:55-56 / train_SSD_L.py:42
    for name in meh_names:
        RemoveParamFromOptim(optimizer, model.module, name)
    runner = build_runner(cfg.runner, default_args=dict(model=model, optimizer=optimizer, work_dir=cfg.work_dir, logger=logger, meta=meta))
    meh_params = [p for n in meh_names for p in getattr(head, n).parameters()]
    runner.optimizer_L = FusedSGD(meh_params, lr=cfg.optimizer.lr,
                                  momentum=cfg.optimizer.momentum, weight_decay=cfg.optimizer.weight_decay)
    runner.timestamp = timestamp
    runner.register_training_hooks(cfg.lr_config, cfg.optimizer_config, cfg.checkpoint_config, cfg.log_config, cfg.get('momentum_config', None))
    for i, hook in enumerate(runner.hooks):
        if type(hook).__name__ == 'OptimizerHook':
            runner.hooks.pop(i)
            break
    if validate:                      # train_Lambda.py:60-70
        from .. import datasets as _ds
        from ..mmcv_lite import EvalHook
        val_cfg = dict(cfg.data.val)
        val_samples_per_gpu = val_cfg.pop('samples_per_gpu', 1)
        val_dataset = _ds.build_dataset(val_cfg, dict(test_mode=True))
        val_dataloader = _ds.build_dataloader(val_dataset, samples_per_gpu=val_samples_per_gpu, workers_per_gpu=cfg.data.workers_per_gpu,
                                          dist=distributed, shuffle=False)
        eval_cfg = dict(cfg.get('evaluation', {}))
        eval_cfg['by_epoch'] = cfg.runner['type'] != 'IterBasedRunner'
        runner.register_hook(EvalHook(val_dataloader, **eval_cfg))
    if cfg.get('resume_from'):
        runner.resume(cfg.resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    return runner.run_SSL(data_loaders, cfg.workflow, cfg.total_epochs, onlyEval=cfg.get('onlyEval', False))
