"""Fixtures for the real VOC data path from the REFERENCE's own classes, imported from /root/reference under
tools/golden/mmcv_shim.py: XMLDataset / VOCDataset parsing + filtering + aspect-ratio flags (mmdet/datasets/xml_style.py:13-120,
custom.py:163-221, voc.py), GroupSampler / DistributedGroupSampler index orders (samplers/group_sampler.py:10-148), and the geometry of
the pipeline transforms -- Resize scale draws and box scaling, RandomFlip decisions and bbox_flip, Expand, MinIoURandomCrop, Pad, the
meta keys Collect hands on (pipelines/transforms.py:26-722, formating.py:290-318) -- on the tiny VOC tree of tests/synth.write_tiny_voc.

The five `mmcv.im*` pixel functions (mmcv is absent) are numpy / PIL stand-ins defined below: a fixture therefore holds GEOMETRY and random
draw order only (shapes, scale factors, boxes, labels, indices), never pixel values.  `imrescale` restates mmcv 1.3.8's published size
rule new = int(old * scale + 0.5).

    python tools/golden/make_golden_data.py     ->  tests/golden/voc_data.npz
"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')
import mmcv_shim  # noqa: E402

mmcv_shim.install()
import mmcv  # noqa: E402   (the shim's stub module: give it the functions the data path calls)
from PIL import Image  # noqa: E402


def _rescale_size(old_size, scale, return_scale=False):
    w, h = old_size
    if isinstance(scale, (float, int)):
        sf = scale
    else:
        max_long, max_short = max(scale), min(scale)
        sf = min(max_long / max(h, w), max_short / min(h, w))
    new = (int(w * float(sf) + 0.5), int(h * float(sf) + 0.5))
    return (new, sf) if return_scale else new


def _imresize(img, size, return_scale=False, interpolation='bilinear', out=None, backend=None):
    h, w = img.shape[:2]
    arr = np.asarray(Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).resize(size, Image.BILINEAR)).astype(img.dtype)
    return (arr, size[0] / w, size[1] / h) if return_scale else arr


def _imrescale(img, scale, return_scale=False, interpolation='bilinear', backend=None):
    h, w = img.shape[:2]
    new, sf = _rescale_size((w, h), scale, return_scale=True)
    out = _imresize(img, new)
    return (out, sf) if return_scale else out


def _imflip(img, direction='horizontal'):
    return {'horizontal': np.flip(img, 1), 'vertical': np.flip(img, 0), 'diagonal': np.flip(img, (0, 1))}[direction]


def _impad(img, *, shape=None, padding=None, pad_val=0, padding_mode='constant'):
    out = np.full(tuple(shape[:2]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def _impad_to_multiple(img, divisor, pad_val=0):
    return _impad(img, shape=(int(np.ceil(img.shape[0] / divisor)) * divisor, int(np.ceil(img.shape[1] / divisor)) * divisor), pad_val=pad_val)


def _imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    return (img - mean) / std


class _DC:          # mmcv.parallel.DataContainer
    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self.data, self.stack, self.cpu_only = data, stack, cpu_only


for name, fn in dict(list_from_file=lambda f, prefix='', offset=0, max_num=0: [prefix + l.rstrip('\n\r') for l in open(f)],
                     is_list_of=lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq), is_str=lambda x: isinstance(x, str),
                     imrescale=_imrescale, imresize=_imresize, imflip=_imflip, impad=_impad, impad_to_multiple=_impad_to_multiple,
                     imnormalize=_imnormalize, rescale_size=_rescale_size).items():
    setattr(mmcv, name, fn)
import mmcv.parallel as _par  # noqa: E402

_par.DataContainer = _DC
import mmcv.runner as _mr  # noqa: E402

_mr.get_dist_info = lambda: (0, 1)
import mmcv.utils as _mu  # noqa: E402

_mu.print_log = lambda *a, **k: None

from mmdet.datasets.pipelines import transforms as T  # noqa: E402
from mmdet.datasets.pipelines.formating import Collect  # noqa: E402
from mmdet.datasets.samplers.group_sampler import DistributedGroupSampler, GroupSampler  # noqa: E402
from mmdet.datasets.voc import VOCDataset  # noqa: E402

from tests import synth  # noqa: E402

out = {}
tmp = tempfile.mkdtemp()
voc = synth.write_tiny_voc(os.path.join(tmp, 'VOCdevkit', 'VOC2007'))

# ---------------------------------------------------------------- XML parsing, filtering, flags
ds = VOCDataset(ann_file=voc + 'ImageSets/Main/trainval.txt', pipeline=[], img_prefix=voc)
out['ids'] = np.array([d['id'] for d in ds.data_infos])
out['wh'] = np.array([[d['width'], d['height']] for d in ds.data_infos])
out['flag'] = np.asarray(ds.flag)
out['year'] = np.int64(ds.year)
for i in range(len(ds)):
    a = ds.get_ann_info(i)
    for k in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore'):
        out[f'ann{i}_{k}'] = a[k]
    out[f'cat_ids{i}'] = np.array(ds.get_cat_ids(i))
dt = VOCDataset(ann_file=voc + 'ImageSets/Main/trainval.txt', pipeline=[], img_prefix=voc, test_mode=True)
out['test_ids'] = np.array([d['id'] for d in dt.data_infos])
dm = VOCDataset(ann_file=voc + 'ImageSets/Main/trainval.txt', pipeline=[], img_prefix=voc, min_size=120)     # drops boxes narrower than 120
out['minsize_ann0_bboxes'] = dm.get_ann_info(0)['bboxes']
out['minsize_ann0_ignore'] = dm.get_ann_info(0)['bboxes_ignore']


# ---------------------------------------------------------------- samplers (index orders under numpy / torch seeds)
class _Flags:
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.uint8)

    def __len__(self):
        return len(self.flag)


flags = (np.random.RandomState(3).rand(37) < 0.35).astype(np.uint8)
out['sampler_flags'] = flags
for spg in (2, 4):
    np.random.seed(11)
    out[f'group_sampler_spg{spg}'] = np.array(list(GroupSampler(_Flags(flags), samples_per_gpu=spg)))
for rank in (0, 1):
    for epoch in (0, 3):
        s = DistributedGroupSampler(_Flags(flags), samples_per_gpu=2, num_replicas=2, rank=rank, seed=5)
        s.set_epoch(epoch)
        out[f'dist_sampler_r{rank}_e{epoch}'] = np.array(list(s))
out['dist_sampler_len'] = np.int64(len(DistributedGroupSampler(_Flags(flags), samples_per_gpu=2, num_replicas=2, rank=0, seed=5)))

# ---------------------------------------------------------------- transforms: geometry and random-draw order
boxes = np.array([[47, 239, 194, 370], [7, 11, 351, 297], [480, 5, 499, 60]], np.float32)
labels = np.array([11, 14, 3], np.int64)


def base(h=375, w=500):
    img = np.random.RandomState(1).randint(0, 255, (h, w, 3)).astype(np.float32)
    return dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], bbox_fields=['gt_bboxes'], gt_bboxes=boxes.copy(),
                gt_labels=labels.copy(), filename='x.jpg', ori_filename='x.jpg')


# Resize: keep-ratio (1000, 600), fixed, multiscale 'value' and 'range' draws
r = T.Resize(img_scale=(1000, 600), keep_ratio=True)(base())
out['resize_keep_shape'], out['resize_keep_sf'], out['resize_keep_boxes'] = np.array(r['img_shape']), r['scale_factor'], r['gt_bboxes']
r = T.Resize(img_scale=(300, 300), keep_ratio=False)(base())
out['resize_fix_shape'], out['resize_fix_sf'], out['resize_fix_boxes'] = np.array(r['img_shape']), r['scale_factor'], r['gt_bboxes']
r = T.Resize(img_scale=(1000, 600), keep_ratio=True)(base(500, 333))
out['resize_tall_shape'], out['resize_tall_sf'] = np.array(r['img_shape']), r['scale_factor']
np.random.seed(21)
rs = T.Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='range', keep_ratio=True)
sc = []
for _ in range(8):
    d = {}
    rs._random_scale(d)
    sc.append(d['scale'])
out['resize_range_scales'] = np.array(sc)
np.random.seed(22)
rs = T.Resize(img_scale=[(1333, 640), (1333, 672), (1333, 800)], multiscale_mode='value', keep_ratio=True)
sc = []
for _ in range(8):
    d = {}
    rs._random_scale(d)
    sc.append(list(d['scale']) + [d['scale_idx']])
out['resize_value_scales'] = np.array(sc)

# RandomFlip: decisions under a seed, bbox_flip in the three directions
np.random.seed(23)
fl = T.RandomFlip(flip_ratio=0.5)
dec = []
for _ in range(12):
    r = fl(base())
    dec.append([int(bool(r['flip'])), {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}[r['flip_direction']]])
out['flip_decisions'] = np.array(dec)
np.random.seed(24)
fl3 = T.RandomFlip(flip_ratio=[0.3, 0.2, 0.2], direction=['horizontal', 'vertical', 'diagonal'])
dec, fboxes = [], []
for _ in range(12):
    r = fl3(base())
    dec.append({None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}[r['flip_direction']] if r['flip'] else 0)
    fboxes.append(r['gt_bboxes'])
out['flip3_decisions'], out['flip3_boxes'] = np.array(dec), np.stack(fboxes)
for d_ in ('horizontal', 'vertical', 'diagonal'):
    out[f'bbox_flip_{d_}'] = fl.bbox_flip(boxes.copy(), (375, 500, 3), d_)

# Pad
r = T.Pad(size_divisor=32)(dict(img=np.zeros((600, 800, 3), np.float32), img_fields=['img']))
out['pad_shape'] = np.array(r['pad_shape'])

# Expand (SSD): canvas size and box offsets under a seed
np.random.seed(25)
ex = T.Expand(mean=(123.675, 116.28, 103.53), to_rgb=True, ratio_range=(1, 4))
eshape, eboxes = [], []
for _ in range(10):
    r = ex(base(300, 400))
    eshape.append(r['img'].shape[:2])
    eboxes.append(r['gt_bboxes'])
out['expand_shapes'], out['expand_boxes'] = np.array(eshape), np.stack(eboxes)
out['expand_fill'] = ex(dict(base(300, 400)))['img'][0, 0] if False else np.array(ex.mean, np.float32)

# MinIoURandomCrop (SSD)
np.random.seed(26)
mc = T.MinIoURandomCrop(min_ious=(0.1, 0.3, 0.5, 0.7, 0.9), min_crop_size=0.3)
cshape, cn = [], []
for k in range(12):
    r = mc(base(300, 400))
    cshape.append(r['img'].shape[:2])
    cn.append(len(r['gt_bboxes']))
    out[f'crop{k}_boxes'], out[f'crop{k}_labels'] = r['gt_bboxes'], r['gt_labels']
out['crop_shapes'], out['crop_counts'] = np.array(cshape), np.array(cn)

# the meta keys Collect hands to the model after the whole RetinaNet train pipeline (geometry only)
np.random.seed(27)
res = base()
for tr in (T.Resize(img_scale=(1000, 600), keep_ratio=True), T.RandomFlip(flip_ratio=0.5),
           T.Normalize(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True), T.Pad(size_divisor=32)):
    res = tr(res)
col = Collect(keys=['img', 'gt_bboxes', 'gt_labels'])(res)
meta = col['img_metas'].data
out['meta_keys'] = np.array(sorted(meta.keys()))
out['meta_img_shape'], out['meta_pad_shape'], out['meta_ori_shape'] = np.array(meta['img_shape']), np.array(meta['pad_shape']), np.array(meta['ori_shape'])
out['meta_scale_factor'], out['meta_flip'] = np.asarray(meta['scale_factor']), np.bool_(meta['flip'])
out['meta_boxes'] = res['gt_bboxes']

path = os.path.join(ROOT, 'tests', 'golden', 'voc_data.npz')
np.savez_compressed(path, **out)
print(f'voc_data: {os.path.getsize(path) / 1024:.1f} KB, {len(out)} arrays')
print('ids', out['ids'], 'flag', out['flag'], 'crop shapes', out['crop_shapes'][:4].tolist(), 'expand', out['expand_shapes'][:3].tolist())
