#!/usr/bin/env python
"""Active-learning driver with the reference's CLI and control flow (tools/train_RetinaNet.py:49-255):
seeds = 20, per cycle {build + init model, outer_epoch x train_detector_SSL phases, save, HUA-score the pool, select}.

    python tools/train_RetinaNet.py --work-dir demo --synthetic 64            # synthetic VOC-shaped pool (no dataset needed)
    python -m torch.distributed.run --nproc-per-node 8 tools/train_RetinaNet.py --launcher pytorch ...   # one rank per MI355X
"""
import argparse
import math
import os
import os.path as osp
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from aod_meh_hua_amd.apis import calculate_uncertainty, train_detector_SSL  # noqa: E402
from aod_meh_hua_amd.datasets import build_dataloader, build_dataset  # noqa: E402
from aod_meh_hua_amd.mmcv_lite import Config, MMDataParallel, mkdir_or_exist  # noqa: E402
from aod_meh_hua_amd.models import build_detector  # noqa: E402
from aod_meh_hua_amd.utils import get_root_logger  # noqa: E402
from aod_meh_hua_amd.utils.active_datasets import create_X_L_file, get_X_L_0_prev, update_X_L  # noqa: E402
from aod_meh_hua_amd.utils.functions import DelJunkSave, EditCfg, ResumeCycle  # noqa: E402

# module-level knobs, as in the reference (:28-43)
onlyEval = False
load_cycle = -1
resume_cycle = -1
isSave = True
editCfg = {'uncertainty_pool2': 'objectSum_scaleMax_classSum'}
clsW = False
zeroRate = 0.15
saveMaxConf = False
useMaxConf = 'False'
score_thr = 0.3
iou_thr = 0.9


def parse_args(default_config='configs/_base_/Config_RetinaNet.py', default_size=512):
    base_dir = osp.dirname(osp.dirname(osp.abspath(__file__)))
    p = argparse.ArgumentParser(description='Train a detector (active learning, MEH + HUA)')
    p.add_argument('--config', default=osp.join(base_dir, default_config))
    p.add_argument('--work-dir', default='WORK_DIR')
    p.add_argument('--resume-from')
    p.add_argument('--load-from')
    p.add_argument('--bbox-head')
    p.add_argument('--uncertainty', help='uncertainty type (accepted like the reference accepts it, tools/train_RetinaNet.py:56: never read there either)')
    p.add_argument('--no-validate', default=False, help='whether not to evaluate during training')
    group_gpus = p.add_mutually_exclusive_group()
    group_gpus.add_argument('--gpus', type=int, default=None, help='number of gpus to use (only applicable to non-distributed training)')
    group_gpus.add_argument('--gpu-ids', type=int, default=None, nargs='+', help='ids of gpus to use (only applicable to non-distributed training)')
    p.add_argument('--deterministic', action='store_true',
                   help='the reference sets cuDNN to deterministic here; the HIP kernels reduce in a fixed order already (slab-ordered weight '
                        'gradients, ordered split-K) except the fp32-atomic column sums of bias / BN-shift gradients -- the flag switches those '
                        'to ordered partial sums as well (aod_set_deterministic: two runs from the same state are then bit-identical; ~1 %% slower)')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none', help='job launcher')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--Unc-type', type=str)
    p.add_argument('--precision', choices=['bf16x3', 'bf16'], default=None,
                   help='arithmetic of the conv stack (default: AOD_CONV_PREC or bf16x3): bf16x3 = reference precision, bf16 = fast mode')
    p.add_argument('--synthetic', type=int, default=0, help='run on a synthetic VOC-shaped pool of this many images')
    p.add_argument('--synthetic-size', type=int, default=default_size)
    p.add_argument('--cycles', type=int, default=None, help='override the number of AL cycles')
    p.add_argument('--samples-per-gpu', type=int, default=None)
    args = p.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    return args


def init_dist(launcher, backend='nccl', **kwargs):
    """mmcv.runner.init_dist as the reference calls it (tools/train_RetinaNet.py:119-121): one process per GPU; 'nccl' is RCCL over xGMI on
    ROCm.  pytorch: the environment of torch.distributed.run; slurm: rank / size from SLURM_PROCID / SLURM_NTASKS, the first host of
    SLURM_NODELIST as the rendezvous address (mmcv: `scontrol show hostname`), local rank = rank modulo the visible devices; mpi: the
    OMPI_COMM_WORLD_* variables."""
    import subprocess
    import torch.distributed as dist
    if launcher == 'slurm':
        rank, world = int(os.environ['SLURM_PROCID']), int(os.environ['SLURM_NTASKS'])
        if 'MASTER_ADDR' not in os.environ:
            try:
                os.environ['MASTER_ADDR'] = subprocess.getoutput(f"scontrol show hostname {os.environ['SLURM_NODELIST']} | head -n1").strip() or '127.0.0.1'
            except KeyError:
                os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ.setdefault('MASTER_PORT', str(kwargs.pop('port', 29500)))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank % max(torch.cuda.device_count(), 1)))
    elif launcher == 'mpi':
        os.environ.update(RANK=os.environ['OMPI_COMM_WORLD_RANK'], WORLD_SIZE=os.environ['OMPI_COMM_WORLD_SIZE'],
                          LOCAL_RANK=os.environ.get('OMPI_COMM_WORLD_LOCAL_RANK', '0'))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(kwargs.pop('port', 29500)))
    elif launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    kwargs.pop('port', None)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist.init_process_group(backend=backend, **kwargs)


def main(default_config='configs/_base_/Config_RetinaNet.py', default_size=512):
    args = parse_args(default_config, default_size)
    cfg = Config.fromfile(args.config)
    seed = 20
    torch.manual_seed(seed), np.random.seed(seed), random.seed(seed)
    cfg.seed, cfg.onlyEval = seed, onlyEval
    if args.bbox_head:
        cfg.model.bbox_head.type = args.bbox_head
    str2unc = {'SACA': 'scaleAvg_classAvg', 'SSCS': 'scaleSum_classSum', 'SACS': 'scaleAvg_classSum', 'SSCA': 'scaleSum_classAvg'}
    if args.Unc_type:
        cfg.uncertainty_pool2 = str2unc[args.Unc_type]
    base_dir = osp.dirname(osp.dirname(osp.abspath(__file__)))
    cfg.work_dir = osp.join(base_dir, 'work_dirs', args.work_dir)
    mkdir_or_exist(cfg.work_dir)
    cfg.save_dir = osp.join(cfg.work_dir, 'model_save')
    mkdir_or_exist(cfg.save_dir)
    # train_RetinaNet.py:107-110: --gpu-ids wins, else range(--gpus) (default: one device)
    cfg.gpu_ids = args.gpu_ids if args.gpu_ids is not None else list(range(args.gpus or 1))
    if args.precision:
        from aod_meh_hua_amd import functional as AF
        AF.set_precision(args.precision)
    distributed = args.launcher != 'none'
    if distributed:
        init_dist(args.launcher, **cfg.dist_params)                          # (sets RANK / LOCAL_RANK / WORLD_SIZE for slurm and mpi)
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if distributed:
        import torch.distributed as dist
        cfg.gpu_ids = list(range(dist.get_world_size()))
    rank = int(os.environ.get('RANK', 0))
    if editCfg:
        EditCfg(cfg, editCfg)
    cfg.model.backbone.pop('init_cfg', None) if args.synthetic else None
    if args.synthetic:            # the pool is N deterministic synthetic images
        n = args.synthetic
        ann = osp.join(cfg.work_dir, 'synthetic_all.txt')
        if rank == 0:
            np.savetxt(ann, np.array([f'synthetic_{i}' for i in range(n)]), fmt='%s')
        if distributed:
            torch.distributed.barrier()
        ds = dict(type='SyntheticVOCDataset', size=(args.synthetic_size, args.synthetic_size), ann_file=[ann])
        cfg.data.train = dict(type='RepeatDataset', times=cfg.X_L_repeat, dataset=dict(ds))
        cfg.data.test = dict(ds)
        cfg.data.val = dict(ds, ann_file=None, indices=list(range(min(n, 8))))        # (no VOC2007 test split offline: a few pool images)
        cfg.X_L_0_size, cfg.X_S_size = max(n // 8, 1), max(n // 16, 1)
    if args.cycles is not None:
        cfg.cycles = list(range(args.cycles))
    if args.samples_per_gpu:
        cfg.data.samples_per_gpu = args.samples_per_gpu
    cfg.dump(osp.join(cfg.work_dir, osp.basename(args.config)))
    timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
    logger = get_root_logger(log_file=osp.join(cfg.work_dir, f'{timestamp}.log'), log_level=cfg.log_level)
    meta = dict(exp_name=osp.basename(args.config))
    if args.deterministic:
        from aod_meh_hua_amd import functional as _AF
        _AF.set_deterministic(True)
        logger.info('--deterministic: weight gradients and split-K sums are reduced in a fixed order by construction; bias / BN-shift column sums '
                    'now as ordered partial sums too (aod_set_deterministic)')
    # train_RetinaNet.py:139-141: 8 loader workers when started from a shell, else 0; here: 8 for real image data (AOD_WORKERS overrides), 0 for
    # the synthetic pool (its samples are generated in-process)
    cfg.data.workers_per_gpu = 0 if args.synthetic else int(os.environ.get('AOD_WORKERS', 8))

    X_L, X_U, X_all, all_anns = get_X_L_0_prev(cfg)
    if rank == 0:
        np.save(cfg.work_dir + '/X_L_0.npy', X_L), np.save(cfg.work_dir + '/X_U_0.npy', X_U)
    notResumed = True
    for cycle in cfg.cycles:
        if resume_cycle >= 0 and notResumed:
            X_L, X_U = ResumeCycle(cfg, cycle, resume_cycle)
            if not isinstance(X_L, np.ndarray):
                continue
            notResumed = False
        logger.info(f'Current cycle is {cycle} cycle.  len of X_U:{len(X_U)}, X_L:{len(X_L)}')
        cfg = create_X_L_file(cfg, X_L, all_anns, cycle)
        model = build_detector(cfg.model)
        model.init_weights()
        if cfg.model.train_cfg.get('bias') == 'uniform':                     # :158-162
            head = model.bbox_head
            if hasattr(head, 'retina_cls'):
                N, k = head.num_anchors, head.retina_cls.bias.numel()
                torch.nn.init.uniform_(head.retina_cls.bias, -math.sqrt(1 / (N * k)), math.sqrt(1 / (N * k)))
            else:                                                             # tools/train_SSD.py:172-178
                for conv in head.cls_convs:
                    torch.nn.init.uniform_(conv[-1].bias, -math.sqrt(1 / 2000), math.sqrt(1 / 2000))
        if load_cycle >= 0:
            from aod_meh_hua_amd.mmcv_lite import load_checkpoint
            cfg_name = osp.splitext(osp.basename(args.config))[0]
            load_checkpoint(model, f'{cfg.save_dir}/{cfg_name}_Cycle{load_cycle}_Epoch{cfg.runner.max_epochs}_mycode.pth')
        datasets = [build_dataset(cfg.data.train)]
        model.CLASSES = datasets[0].CLASSES
        validate = not args.no_validate                                       # :60,193,210 (the reference evaluates unless told not to)
        for epoch in range(cfg.outer_epoch):
            cfg.lr_config.step = [1000]
            if epoch != cfg.outer_epoch - 1:                                  # :179-183
                cfg.evaluation.interval = 100
            cfg.optimizer['lr'] = 0.001
            if epoch == 0:
                logger.info(f'Epoch = {epoch}, First Label Set Training')
                cfg.total_epochs = cfg.epoch_ratio[0]
                train_detector_SSL(model, [build_dataset(cfg.data.train)], cfg, distributed=distributed, validate=validate, timestamp=timestamp, meta=meta)
            cfg.evaluation.interval = 100                                     # :198-201: evaluate only at the end of the last outer epoch
            if epoch == cfg.outer_epoch - 1:
                cfg.lr_config.step = [2]
                cfg.evaluation.interval = cfg.epoch_ratio[0]
            logger.info(f'Epoch = {epoch}, Fully-Supervised Learning')
            cfg.total_epochs = cfg.epoch_ratio[0]
            train_detector_SSL(model, [build_dataset(cfg.data.train)], cfg, distributed=distributed, validate=validate, timestamp=timestamp, meta=meta)
        if isSave and rank == 0:
            for f in os.listdir(cfg.save_dir):
                if '_mycode' not in f:
                    os.remove(osp.join(cfg.save_dir, f))
            cfg_name = osp.splitext(osp.basename(args.config))[0]
            torch.save(model.state_dict(), f'{cfg.save_dir}/{cfg_name}_Cycle{cycle}_Epoch{cfg.runner.max_epochs}_mycode.pth')
        if cycle != cfg.cycles[-1]:
            dataset_al = build_dataset(cfg.data.test)
            data_loader = build_dataloader(dataset_al, samples_per_gpu=cfg.data.samples_per_gpu, workers_per_gpu=cfg.data.workers_per_gpu,
                                           dist=False, shuffle=False)
            poolModel = MMDataParallel(model, device_ids=cfg.gpu_ids)
            with torch.no_grad():
                uncertainty = calculate_uncertainty(cfg, poolModel, data_loader, return_box=False, showNMS=False, saveUnc=False,
                                                    saveMaxConf=saveMaxConf, clsW=clsW, scaleUnc=False, score_thr=score_thr, iou_thr=iou_thr)
            maxconf = None
            if saveMaxConf:                                                   # :236-240
                uncertainty, maxconf = uncertainty
                maxconf = maxconf.cpu().numpy() if torch.is_tensor(maxconf) else np.asarray(maxconf)
            uncertainty = uncertainty.cpu().numpy() if torch.is_tensor(uncertainty) else np.asarray(uncertainty)
            X_L, X_U = update_X_L(uncertainty, X_all, X_L, cfg.X_S_size, zeroRate=zeroRate, maxconf=maxconf, useMaxConf=useMaxConf)
            if rank == 0:
                np.save(cfg.work_dir + f'/X_L_{cycle + 1}.npy', X_L), np.save(cfg.work_dir + f'/X_U_{cycle + 1}.npy', X_U)
                np.save(cfg.work_dir + f'/Unc_{cycle + 1}.npy', uncertainty)
            logger.info(f'cycle {cycle}: scored {len(uncertainty)} images, {int((uncertainty > 0).sum())} non-zero, selected {cfg.X_S_size}')
        if rank == 0:
            DelJunkSave(cfg.work_dir)


if __name__ == '__main__':
    main()
