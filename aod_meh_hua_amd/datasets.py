"""Data side of the boundary (SURVEY 8b "Data contract"): the synthetic VOC-shaped dataset the benchmarks and the AL-loop plumbing use
 and the real VOC path: XMLDataset / VOCDataset (datasets/xml_style.py:13-170, voc.py:11-94,
custom.py:56-363), ConcatDataset / RepeatDataset (dataset_wrappers.py), GroupSampler (samplers/group_sampler.py:9-50), mmcv collate
with padding, build_dataset / build_dataloader (builder.py:40-147).  Transforms live in pipelines.py."""
import os.path as osp
import xml.etree.ElementTree as ET

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
class DevicePhiloxPool(Dataset):
    """Unlabeled pool whose images are generated ON the device ('pool of 10 000 synthetic 512^2 images generated on-device
    from Philox(seed=20, image_id)'): image i is a pure function of (seed, i), so any sharding over ranks / batches scores the same pool.
    `device_batch(idxs, device)` is the hook apis/test.py single_gpu_uncertainty uses instead of host collate + H2D copy; `__getitem__`
    gives the same image on the host for small cross-checks."""
    CLASSES = VOC_CLASSES

    def __init__(self, num_images=10000, size=(512, 512), seed=20, **kw):
        self.num_images, self.size, self.seed = int(num_images), tuple(size), int(seed)
        self.flag = np.zeros(self.num_images, dtype=np.uint8)
        self._buf = {}

    def __len__(self):
        return self.num_images

    def _meta(self, idx):
        H, W = self.size
        return dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=np.ones(4, np.float32), flip=False,
                    flip_direction=None, filename=f'philox_{idx}', ori_filename=f'philox_{idx}', image_id=idx)

    def device_batch(self, idxs, device, image_ids=None, out=None):
        """-> dict(img=[tensor [B,3,H,W] on device], img_metas=[list of dicts]).  The image tensor is `out` when given (the scoring graph's
        input buffer, graphs.GraphedScore.static_image: nothing is copied afterwards), else a per-batch-size static buffer; one kernel launch."""
        from . import _C
        H, W = self.size
        B = len(idxs)
        if out is not None and tuple(out.shape) == (B, 3, H, W) and out.dtype == torch.float32 and out.is_contiguous():
            img = out
        else:
            key = (B, str(device))
            if key not in self._buf:
                self._buf[key] = torch.empty(B, 3, H, W, device=device)
            img = self._buf[key]
        ids = image_ids if image_ids is not None else torch.tensor(list(idxs), dtype=torch.int64).to(device, non_blocking=True)
        _C.call('aod_synth_normal_images', _C.ptr(img), B, 3 * H * W, self.seed, _C.ptr(ids), _C.stream())
        return dict(img=[img], img_metas=[[self._meta(int(i)) for i in idxs]])

    def __getitem__(self, i):
        dev = torch.device('cuda', torch.cuda.current_device())
        d = self.device_batch([int(i)], dev)
        return dict(img=d['img'][0][0].cpu(), img_metas=d['img_metas'][0][0])


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


class CustomDataset(Dataset):
    """custom.py:56-363 (detection subset): annotation list -> filtered infos -> aspect-ratio group flag -> pipeline."""
    CLASSES = None

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix='', seg_prefix=None, proposal_file=None,
                 test_mode=False, filter_empty_gt=True):
        from .pipelines import Compose
        self.ann_file, self.data_root, self.img_prefix = ann_file, data_root, img_prefix
        self.test_mode, self.filter_empty_gt, self.proposals = test_mode, filter_empty_gt, None
        self.CLASSES = self.get_classes(classes)
        if self.data_root is not None:
            if not osp.isabs(self.ann_file):
                self.ann_file = osp.join(self.data_root, self.ann_file)
            if not (self.img_prefix is None or osp.isabs(self.img_prefix)):
                self.img_prefix = osp.join(self.data_root, self.img_prefix)
        self.data_infos = self.load_annotations(self.ann_file)
        if not test_mode:
            valid_inds = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in valid_inds]
            self._set_group_flag()
        else:
            self.flag = np.zeros(len(self.data_infos), dtype=np.uint8)
        self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self.data_infos)

    @classmethod
    def get_classes(cls, classes=None):
        if classes is None:
            return cls.CLASSES
        if isinstance(classes, str):
            return [ln.strip() for ln in open(classes) if ln.strip()]
        if isinstance(classes, (tuple, list)):
            return classes
        raise ValueError(f'Unsupported type {type(classes)} of classes.')

    def pre_pipeline(self, results):
        results['img_prefix'], results['seg_prefix'], results['proposal_file'] = self.img_prefix, None, None
        results['bbox_fields'], results['mask_fields'], results['seg_fields'] = [], [], []

    def _filter_imgs(self, min_size=32):
        return [i for i, info in enumerate(self.data_infos) if min(info['width'], info['height']) >= min_size]

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i in range(len(self)):
            info = self.data_infos[i]
            if info['width'] / info['height'] > 1:
                self.flag[i] = 1

    def _rand_another(self, idx):
        return np.random.choice(np.where(self.flag == self.flag[idx])[0])

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    def prepare_train_img(self, idx):
        results = dict(img_info=self.data_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.data_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)


@DATASETS.register_module()
class XMLDataset(CustomDataset):
    """xml_style.py:13-170: PASCAL-VOC style `ImageSets/Main/*.txt` id list + one `Annotations/<id>.xml` per image."""

    def __init__(self, min_size=None, **kwargs):
        assert self.CLASSES or kwargs.get('classes', None), 'CLASSES in `XMLDataset` can not be None.'
        self.min_size = min_size
        super().__init__(**kwargs)
        self.cat2label = {cat: i for i, cat in enumerate(self.CLASSES)}

    def _xml(self, img_id):
        return ET.parse(osp.join(self.img_prefix, 'Annotations', f'{img_id}.xml')).getroot()

    def load_annotations(self, ann_file):
        data_infos = []
        for img_id in [ln.strip() for ln in open(ann_file) if ln.strip()]:
            size = self._xml(img_id).find('size')
            if size is not None:
                width, height = int(size.find('width').text), int(size.find('height').text)
            else:
                from PIL import Image
                with Image.open(osp.join(self.img_prefix, 'JPEGImages', f'{img_id}.jpg')) as im:
                    width, height = im.size
            data_infos.append(dict(id=img_id, filename=f'JPEGImages/{img_id}.jpg', width=width, height=height))
        return data_infos

    def _filter_imgs(self, min_size=32):
        valid = []
        for i, info in enumerate(self.data_infos):
            if min(info['width'], info['height']) < min_size:
                continue
            if self.filter_empty_gt:
                if any(obj.find('name').text in self.CLASSES for obj in self._xml(info['id']).findall('object')):
                    valid.append(i)
            else:
                valid.append(i)
        return valid

    def get_ann_info(self, idx):
        """:95-160: 1-based VOC coordinates -> 0-based (bbox - 1); `difficult` objects and (min_size) tiny ones go to the ignore set."""
        cat2label = {cat: i for i, cat in enumerate(self.CLASSES)}
        bboxes, labels, bboxes_ignore, labels_ignore = [], [], [], []
        for obj in self._xml(self.data_infos[idx]['id']).findall('object'):
            name = obj.find('name').text
            if name not in self.CLASSES:
                continue
            label = cat2label[name]
            difficult = obj.find('difficult')
            difficult = 0 if difficult is None else int(difficult.text)
            bb = obj.find('bndbox')
            bbox = [int(float(bb.find(k).text)) for k in ('xmin', 'ymin', 'xmax', 'ymax')]
            ignore = False
            if self.min_size:
                assert not self.test_mode
                if bbox[2] - bbox[0] < self.min_size or bbox[3] - bbox[1] < self.min_size:
                    ignore = True
            if difficult or ignore:
                bboxes_ignore.append(bbox), labels_ignore.append(label)
            else:
                bboxes.append(bbox), labels.append(label)
        bboxes = np.array(bboxes, ndmin=2) - 1 if bboxes else np.zeros((0, 4))
        labels = np.array(labels) if labels else np.zeros((0,))
        bboxes_ignore = np.array(bboxes_ignore, ndmin=2) - 1 if bboxes_ignore else np.zeros((0, 4))
        labels_ignore = np.array(labels_ignore) if labels_ignore else np.zeros((0,))
        return dict(bboxes=bboxes.astype(np.float32), labels=labels.astype(np.int64), bboxes_ignore=bboxes_ignore.astype(np.float32),
                    labels_ignore=labels_ignore.astype(np.int64))

    def get_cat_ids(self, idx):
        cat2label = {cat: i for i, cat in enumerate(self.CLASSES)}
        return [cat2label[o.find('name').text] for o in self._xml(self.data_infos[idx]['id']).findall('object')
                if o.find('name').text in self.CLASSES]


@DATASETS.register_module()
class VOCDataset(XMLDataset):
    """voc.py:11-94."""
    CLASSES = VOC_CLASSES

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if 'VOC2007' in self.img_prefix:
            self.year = 2007
        elif 'VOC2012' in self.img_prefix:
            self.year = 2012
        else:
            raise ValueError('Cannot infer dataset year from img_prefix')

    def evaluate(self, results, metric='mAP', logger=None, proposal_nums=(100, 300, 1000), iou_thr=0.5, scale_ranges=None, **kwargs):
        from .core.evaluation import evaluate_voc
        return evaluate_voc(results, [self.get_ann_info(i) for i in range(len(self))], year=self.year, classes=self.CLASSES, metric=metric,
                            logger=logger, iou_thr=iou_thr)


class ConcatDataset(torch.utils.data.ConcatDataset):
    """dataset_wrappers.py:12-125 (VOC07+12 trainval): concatenated flags; evaluation is per-dataset in the reference and unused here."""

    def __init__(self, datasets, separate_eval=True):
        super().__init__(datasets)
        self.CLASSES = datasets[0].CLASSES
        self.flag = np.concatenate([d.flag for d in datasets]) if hasattr(datasets[0], 'flag') else None

    def get_ann_info(self, idx):
        import bisect
        d = bisect.bisect_right(self.cumulative_sizes, idx)
        return self.datasets[d].get_ann_info(idx if d == 0 else idx - self.cumulative_sizes[d - 1])


def build_dataset(cfg, default_args=None):
    """builder.py:40-73: list of cfgs or list-valued ann_file -> ConcatDataset; RepeatDataset wrapper; registry otherwise."""
    if isinstance(cfg, (list, tuple)):
        return ConcatDataset([build_dataset(c, default_args) for c in cfg])
    if cfg['type'] == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    if cfg['type'] != 'SyntheticVOCDataset' and isinstance(cfg.get('ann_file'), (list, tuple)):
        ann_files, prefixes = cfg['ann_file'], cfg.get('img_prefix')
        parts = []
        for i, af in enumerate(ann_files):
            c = dict(cfg)
            c['ann_file'] = af
            if isinstance(prefixes, (list, tuple)):
                c['img_prefix'] = prefixes[i]
            parts.append(build_dataset(c, default_args))
        return ConcatDataset(parts)
    return build_from_cfg(cfg, DATASETS, default_args)


class GroupSampler(torch.utils.data.Sampler):
    """samplers/group_sampler.py:9-50: batches never mix the two aspect-ratio groups; each group is padded (by repetition) to a multiple
    of samples_per_gpu, shuffled inside the group, and the batches of both groups are shuffled together (np.random like the reference)."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, 'flag')
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = dataset.flag.astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(s / samples_per_gpu)) * samples_per_gpu for s in self.group_sizes)

    def __iter__(self):
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            indice = np.where(self.flag == i)[0]
            np.random.shuffle(indice)
            num_extra = int(np.ceil(size / self.samples_per_gpu)) * self.samples_per_gpu - len(indice)
            indice = np.concatenate([indice, np.random.choice(indice, num_extra)])
            indices.append(indice)
        indices = np.concatenate(indices)
        indices = [indices[i * self.samples_per_gpu:(i + 1) * self.samples_per_gpu]
                   for i in np.random.permutation(range(len(indices) // self.samples_per_gpu))]
        return iter(np.concatenate(indices).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(torch.utils.data.Sampler):
    """mmdet/datasets/samplers/group_sampler.py:53-148: per epoch (seed + epoch) every aspect-ratio group is shuffled and padded to a
    multiple of samples_per_gpu * world, whole batches are shuffled, and rank r takes the r-th contiguous share -- so a batch never mixes
    portrait and landscape images and every rank draws a different, epoch-dependent share.  `set_epoch` is called by the runner at the
    start of every training epoch (mmcv's DistSamplerSeedHook)."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        import math
        import torch.distributed as tdist
        if num_replicas is None or rank is None:
            ok = tdist.is_available() and tdist.is_initialized()
            num_replicas = tdist.get_world_size() if ok else 1
            rank = tdist.get_rank() if ok else 0
        self.dataset, self.samples_per_gpu, self.num_replicas, self.rank, self.seed, self.epoch = dataset, samples_per_gpu, num_replicas, rank, seed or 0, 0
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(math.ceil(sz * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu for sz in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        import math
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                indice = np.where(self.flag == i)[0]
                indice = indice[list(torch.randperm(int(size), generator=g).numpy())].tolist()
                extra = int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu * self.num_replicas - len(indice)
                tmp = indice.copy()
                for _ in range(extra // size):
                    indice.extend(tmp)
                indice.extend(tmp[:extra % size])
                indices.extend(indice)
        assert len(indices) == self.total_size
        indices = [indices[j] for i in list(torch.randperm(len(indices) // self.samples_per_gpu, generator=g))
                   for j in range(i * self.samples_per_gpu, (i + 1) * self.samples_per_gpu)]
        offset = self.num_samples * self.rank
        return iter(indices[offset:offset + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


def _collate_dc(items, samples_per_gpu):
    """mmcv.parallel.collate for a list of DataContainers (one GPU per process: a single chunk)."""
    first = items[0]
    if first.cpu_only:
        return DataContainer([[d.data for d in items]], first.stack, cpu_only=True)
    if first.stack:
        # pad the last `pad_dims` = 2 dims to the batch maximum (bottom / right, like F.pad in mmcv), then stack
        h = max(d.data.shape[-2] for d in items)
        w = max(d.data.shape[-1] for d in items)
        out = []
        for d in items:
            t = d.data
            out.append(torch.nn.functional.pad(t, (0, w - t.shape[-1], 0, h - t.shape[-2]), value=0))
        return DataContainer([torch.stack(out)], True)
    return DataContainer([[d.data for d in items]], False)


def collate(batch, samples_per_gpu=1):
    """mmcv.parallel.collate: pipeline samples (dicts of DataContainers; test mode: lists per augmentation) or the plain tensors of
    SyntheticVOCDataset -> one dict of DataContainers with a single per-device chunk."""
    first = batch[0]
    if isinstance(first.get('img'), DataContainer):
        return {k: _collate_dc([b[k] for b in batch], samples_per_gpu) for k in first}
    if isinstance(first.get('img'), (list, tuple)):              # MultiScaleFlipAug: [aug][...] -> list over augmentations
        n_aug = len(first['img'])
        out = {}
        for k in first:
            # plain tensors (ImageToTensor): mmcv would refuse unequal shapes (its configs switch to DefaultFormatBundle for
            # samples_per_gpu > 1, train_Lambda.py:61-63); pad bottom / right like a stacked DataContainer instead
            out[k] = [(_collate_dc([b[k][a] for b in batch], samples_per_gpu) if isinstance(first[k][a], DataContainer)
                       else _collate_dc([DataContainer(b[k][a], stack=True) for b in batch], samples_per_gpu).data[0]) for a in range(n_aug)]
        return out
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
        if shuffle and getattr(dataset, 'flag', None) is not None:
            sampler = DistributedGroupSampler(dataset, samples_per_gpu, seed=seed or 0)          # builder.py:105-108
        else:
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle, seed=seed or 0)
        shuffle = False
    if sampler is None and shuffle and getattr(dataset, 'flag', None) is not None and len(np.unique(dataset.flag)) > 1:
        sampler, shuffle = GroupSampler(dataset, samples_per_gpu), False          # builder.py:113-115 (aspect-ratio groups)
    g = torch.Generator()
    g.manual_seed(seed or 0)
    return DataLoader(dataset, batch_size=samples_per_gpu, sampler=sampler, shuffle=shuffle, num_workers=workers_per_gpu,
                      collate_fn=lambda b: collate(b, samples_per_gpu), generator=g, drop_last=False, **kwargs)
