"""Data side of the boundary (SURVEY 8b "Data contract").  Round 1 ships the synthetic VOC-shaped dataset the
benchmarks and the AL-loop plumbing use; the real VOC XML loader + Resize/Flip/Normalize/Pad
pipeline is the first "next" row and is not built yet."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .mmcv_lite import DataContainer, Registry, build_from_cfg

DATASETS = Registry('dataset')
VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse',
               'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


@DATASETS.register_module()
class SyntheticVOCDataset(Dataset):
    """Deterministic per-index samples: img ~ N(0,1) [3,H,W] (already 'normalised'), G ~ U{1..5} boxes with
    w,h ~ U(H/16, 3H/4) clipped inside, labels ~ U{0..19}.  `indices` (or ann_file of integer ids) selects a subset."""
    CLASSES = VOC_CLASSES

    def __init__(self, num_images=64, size=(512, 512), seed=20, indices=None, ann_file=None, test_mode=False, **kw):
        self.size, self.seed, self.test_mode = tuple(size), seed, test_mode
        if ann_file is not None:
            files = ann_file if isinstance(ann_file, (list, tuple)) else [ann_file]
            ids = np.concatenate([np.atleast_1d(np.loadtxt(f, dtype=str)) for f in files]) if len(files) else np.zeros(0, str)
            self.indices = np.array([int(str(i).split('_')[-1]) for i in ids], dtype=np.int64)
        else:
            self.indices = np.arange(num_images) if indices is None else np.asarray(indices)
        self.flag = np.zeros(len(self.indices), dtype=np.uint8)

    def __len__(self):
        return len(self.indices)

    def get_ann_info(self, i):
        """XMLDataset.get_ann_info format (xml_style.py:95-160): numpy boxes / labels (+ empty ignore sets)."""
        d = self[i]
        return dict(bboxes=d['gt_bboxes'].numpy().astype(np.float32), labels=d['gt_labels'].numpy().astype(np.int64),
                    bboxes_ignore=np.zeros((0, 4), np.float32), labels_ignore=np.zeros((0,), np.int64))

    def evaluate(self, results, metric='mAP', logger=None, proposal_nums=(100, 300, 1000), iou_thr=0.5, scale_ranges=None, **kwargs):
        """VOCDataset.evaluate (datasets/voc.py:37-94), VOC07 11-point mode; extra EvalHook kwargs (show, isUnc, out_dir) are ignored
        like the reference's signature swallows them."""
        from .core.evaluation import evaluate_voc
        return evaluate_voc(results, [self.get_ann_info(i) for i in range(len(self))], year=2007, classes=self.CLASSES, metric=metric,
                            logger=logger, iou_thr=iou_thr)

    def __getitem__(self, i):
        idx = int(self.indices[i])
        H, W = self.size
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        img = torch.randn(3, H, W, generator=g)
        G = int(torch.randint(1, 6, (1,), generator=g))
        wh = torch.rand(G, 2, generator=g) * torch.tensor([W * 0.6875, H * 0.6875]) + torch.tensor([W / 16., H / 16.])
        xy = torch.rand(G, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh)
        meta = dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=np.ones(4, np.float32), flip=False,
                    flip_direction=None, filename=f'synthetic_{idx}', ori_filename=f'synthetic_{idx}', image_id=idx)
        return dict(img=img, img_metas=meta, gt_bboxes=torch.cat([xy, xy + wh], 1), gt_labels=torch.randint(0, 20, (G,), generator=g))


@DATASETS.register_module()
class RepeatDataset(Dataset):
    """mmdet/datasets/dataset_wrappers.py:128-170."""

    def __init__(self, dataset, times):
        self.dataset = build_dataset(dataset) if isinstance(dataset, dict) else dataset
        self.times, self.CLASSES = times, self.dataset.CLASSES
        self._ori_len = len(self.dataset)
        self.flag = np.tile(self.dataset.flag, times) if hasattr(self.dataset, 'flag') else None

    def __getitem__(self, idx):
        return self.dataset[idx % self._ori_len]

    def __len__(self):
        return self.times * self._ori_len


def build_dataset(cfg, default_args=None):
    if cfg['type'] == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    if cfg['type'] in ('VOCDataset', 'XMLDataset', 'CocoDataset'):
        raise NotImplementedError('the real VOC/COCO data path is the first "next" row (SURVEY 8f rank 1); '
                                  'use type="SyntheticVOCDataset" (tools/train_RetinaNet.py --synthetic N)')
    return build_from_cfg(cfg, DATASETS, default_args)


def collate(batch, samples_per_gpu=1):
    """mmcv.parallel.collate for this data contract: stack images, keep metas / gts as lists inside DataContainers."""
    return dict(img=DataContainer([torch.stack([b['img'] for b in batch])], stack=True),
                img_metas=DataContainer([[b['img_metas'] for b in batch]], cpu_only=True),
                gt_bboxes=DataContainer([[b['gt_bboxes'] for b in batch]]),
                gt_labels=DataContainer([[b['gt_labels'] for b in batch]]))


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True, seed=None, **kwargs):
    """mmdet/datasets/builder.py:76-139: dist=True -> this rank's strided share of a seeded permutation
    (DistributedGroupSampler semantics, samplers/group_sampler.py:101-142)."""
    import torch.distributed as tdist
    sampler = None
    if dist and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle, seed=seed or 0)
        shuffle = False
    g = torch.Generator()
    g.manual_seed(seed or 0)
    return DataLoader(dataset, batch_size=samples_per_gpu, sampler=sampler, shuffle=shuffle, num_workers=workers_per_gpu,
                      collate_fn=lambda b: collate(b, samples_per_gpu), generator=g, drop_last=False, **kwargs)
