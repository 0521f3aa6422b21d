"""CPU tests of the real VOC data path on a tiny VOC tree written to tmp_path: XML parsing rules, aspect-ratio
groups, every transform of the RetinaNet / SSD pipelines (known answers from the reference formulas), mmcv-style collate, and the
evaluation hook-up.  (cv2 is not available here, so the reference's own transforms cannot be executed for a golden.)"""
import os

import numpy as np
import pytest
import torch

from aod_meh_hua_amd import pipelines as P
from aod_meh_hua_amd.datasets import GroupSampler, build_dataloader, build_dataset, collate
from aod_meh_hua_amd.mmcv_lite import DataContainer

IMG_NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
TRAIN = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True), dict(type='Resize', img_scale=(1000, 600), keep_ratio=True),
         dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', **IMG_NORM), dict(type='Pad', size_divisor=32),
         dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
TEST = [dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(1000, 600), flip=False,
             transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **IMG_NORM),
                         dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]


@pytest.fixture(scope='module')
def voc(tmp_path_factory):
    from tests import synth
    return synth.write_tiny_voc(tmp_path_factory.mktemp('VOCdevkit') / 'VOC2007')


def test_xml_parsing_filtering_and_groups(voc):
    ds = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=TRAIN))
    assert [d['id'] for d in ds.data_infos] == ['000001', '000002', '000005'] and ds.year == 2007     # 3: no VOC object, 4: too small
    assert ds.data_infos[2]['width'] == 400 and ds.data_infos[2]['height'] == 300                      # size read from the JPEG header
    assert ds.flag.tolist() == [1, 0, 1]                                                                # w / h > 1
    a = ds.get_ann_info(0)
    assert a['bboxes'].tolist() == [[47, 239, 194, 370], [7, 11, 351, 297]] and a['labels'].tolist() == [11, 14]   # xml - 1, class ids
    assert a['bboxes_ignore'].tolist() == [[99, 99, 199, 199]] and a['labels_ignore'].tolist() == [7]              # difficult -> ignore
    assert ds.get_ann_info(1)['bboxes'].tolist() == [[9, 19, 299, 479]]                                            # int(float('10.6')) - 1
    test = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=TEST),
                         dict(test_mode=True))
    assert len(test) == 5 and ds.get_cat_ids(0) == [11, 14, 7]
    # list-valued ann_file / img_prefix -> ConcatDataset (VOC07 + VOC12 trainval in the configs)
    cat = build_dataset(dict(type='VOCDataset', ann_file=[voc + 'ImageSets/Main/trainval.txt'] * 2, img_prefix=[voc, voc], pipeline=TRAIN))
    assert len(cat) == 6 and cat.flag.tolist() == [1, 0, 1, 1, 0, 1] and cat.get_ann_info(4)['labels'].tolist() == [6]


def test_train_pipeline_known_answers(voc):
    ds = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=TRAIN))
    np.random.seed(3)
    flips = []
    for idx, (w, h) in ((0, (500, 375)), (1, (333, 500))):
        d = ds[idx]
        meta = d['img_metas'].data
        sf = min(1000 / max(h, w), 600 / min(h, w))
        nw, nh = int(w * sf + 0.5), int(h * sf + 0.5)
        assert meta['ori_shape'] == (h, w, 3) and meta['img_shape'] == (nh, nw, 3)
        assert meta['pad_shape'] == (int(np.ceil(nh / 32)) * 32, int(np.ceil(nw / 32)) * 32, 3)
        assert np.allclose(meta['scale_factor'], [nw / w, nh / h, nw / w, nh / h])
        img = d['img'].data
        assert tuple(img.shape) == (3,) + meta['pad_shape'][:2] and img.dtype == torch.float32
        assert float(img[:, nh:, :].abs().max() if nh < img.shape[1] else 0) == 0 and float(img[:, :, nw:].abs().max() if nw < img.shape[2] else 0) == 0
        gt = ds.get_ann_info(idx)['bboxes'] * meta['scale_factor']
        gt[:, 0::2] = gt[:, 0::2].clip(0, nw)
        gt[:, 1::2] = gt[:, 1::2].clip(0, nh)
        if meta['flip']:
            gt = P.bbox_flip(gt, meta['img_shape'], 'horizontal')
        assert np.allclose(d['gt_bboxes'].data.numpy(), gt, atol=1e-4) and d['gt_labels'].data.dtype == torch.int64
        flips.append(bool(meta['flip']))
    # normalisation: BGR file order -> RGB, (x - mean) / std on the un-flipped, un-resized image
    res = P.Compose([dict(type='LoadImageFromFile'), dict(type='Normalize', **IMG_NORM)])(
        dict(img_info=ds.data_infos[0], img_prefix=voc, bbox_fields=[]))
    from PIL import Image
    rgb = np.asarray(Image.open(voc + 'JPEGImages/000001.jpg').convert('RGB')).astype(np.float32)
    assert np.allclose(res['img'], (rgb - np.float32(IMG_NORM['mean'])) / np.float32(IMG_NORM['std']), atol=1e-5)


def test_transform_unit_rules():
    assert P.rescale_size((500, 375), (1000, 600)) == (800, 600) and P.rescale_size((333, 500), (1000, 600)) == (600, 901)
    assert P.rescale_size((1344, 100), (1333, 800)) == (1333, 99)
    b = np.array([[10., 20., 110., 220.]], np.float32)
    assert P.bbox_flip(b, (300, 400, 3), 'horizontal').tolist() == [[290., 20., 390., 220.]]
    assert P.bbox_flip(b, (300, 400, 3), 'vertical').tolist() == [[10., 80., 110., 280.]]
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    out = P.Pad(size_divisor=4)(dict(img=img, img_fields=['img']))
    assert out['img'].shape == (4, 4, 3) and out['pad_shape'] == (4, 4, 3) and out['img'][2:].sum() == 0 and out['img'][:, 3:].sum() == 0
    r = P.Resize(img_scale=(300, 300), keep_ratio=False)(dict(img=np.zeros((100, 200, 3), np.uint8), img_fields=['img'],
                                                                bbox_fields=['gt_bboxes'], gt_bboxes=np.array([[0, 0, 200, 100.]], np.float32)))
    assert r['img'].shape == (300, 300, 3) and np.allclose(r['scale_factor'], [1.5, 3, 1.5, 3]) and r['gt_bboxes'].tolist() == [[0, 0, 300, 300]]
    np.random.seed(0)
    s = [P.Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='range')._random_scale(d := dict()) or d['scale'] for _ in range(20)]
    assert all(x[0] == 1333 and 640 <= x[1] <= 800 for x in s) and len(set(s)) > 5
    # resize interpolation: constant image stays constant, a horizontal ramp stays a ramp
    ramp = np.tile(np.linspace(0, 255, 64, dtype=np.float32)[None, :, None], (32, 1, 3))
    up = P.imresize(ramp, (128, 64))
    assert up.shape == (64, 128, 3) and np.all(np.diff(up[10, :, 0]) >= -1e-4) and abs(float(up.mean()) - float(ramp.mean())) < 0.5


def test_ssd_augmentations_keep_boxes_consistent():
    np.random.seed(7)
    img = np.random.RandomState(1).uniform(0, 255, (300, 400, 3)).astype(np.float32)
    base = dict(img=img, img_fields=['img'], img_shape=img.shape, bbox_fields=['gt_bboxes'],
                gt_bboxes=np.array([[50, 60, 200, 220], [220, 30, 380, 280]], np.float32), gt_labels=np.array([3, 8]))
    hsv = P.bgr2hsv(img)
    assert np.allclose(P.hsv2bgr(hsv), img, atol=1e-2) and hsv[..., 0].min() >= 0 and hsv[..., 0].max() < 360.001
    for _ in range(20):
        r = P.PhotoMetricDistortion()(dict(base, img=img.copy()))
        assert r['img'].shape == img.shape and r['img'].dtype == np.float32 and np.isfinite(r['img']).all()
    grown = 0
    for _ in range(20):
        r = P.Expand(mean=(123.675, 116.28, 103.53), to_rgb=True, ratio_range=(1, 4))(dict(base, img=img.copy(), gt_bboxes=base['gt_bboxes'].copy()))
        H, W = r['img'].shape[:2]
        if (H, W) != (300, 400):
            grown += 1
            dx, dy = r['gt_bboxes'][0, 0] - 50, r['gt_bboxes'][0, 1] - 60
            assert np.allclose(r['img'][int(dy):int(dy) + 300, int(dx):int(dx) + 400], img) and np.allclose(r['img'][0, 0], [103.53, 116.28, 123.675]) or (dx, dy) == (0, 0)
            assert np.allclose(r['gt_bboxes'][1] - base['gt_bboxes'][1], [dx, dy, dx, dy])
    assert 3 < grown < 18
    cropped = 0
    for _ in range(30):
        r = P.MinIoURandomCrop()(dict(base, img=img.copy(), gt_bboxes=base['gt_bboxes'].copy(), gt_labels=base['gt_labels'].copy()))
        H, W = r['img'].shape[:2]
        assert len(r['gt_bboxes']) == len(r['gt_labels']) >= 1
        assert (r['gt_bboxes'][:, 0::2] >= 0).all() and (r['gt_bboxes'][:, 0::2] <= W).all() and (r['gt_bboxes'][:, 1::2] <= H).all()
        cropped += (H, W) != (300, 400)
    assert cropped > 5


def test_collate_group_sampler_and_loader(voc):
    ds = build_dataset(dict(type='VOCDataset', ann_file=[voc + 'ImageSets/Main/trainval.txt'] * 3, img_prefix=[voc] * 3, pipeline=TRAIN))
    np.random.seed(1)
    order = list(GroupSampler(ds, samples_per_gpu=2))
    assert len(order) == 10 and all(ds.flag[order[i]] == ds.flag[order[i + 1]] for i in range(0, 10, 2))      # 6 + 3(+1 repeat) samples
    dl = build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=0, dist=False, shuffle=True, seed=0)
    batch = next(iter(dl))
    img = batch['img'].data[0]
    metas = batch['img_metas'].data[0]
    assert img.shape[0] == 2 and img.shape[2] == max(m['pad_shape'][0] for m in metas) and img.shape[3] == max(m['pad_shape'][1] for m in metas)
    assert isinstance(batch['gt_bboxes'], DataContainer) and len(batch['gt_bboxes'].data[0]) == 2 and batch['img_metas'].cpu_only
    test = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=TEST), dict(test_mode=True))
    tb = collate([test[0], test[1]])
    assert isinstance(tb['img'], list) and len(tb['img']) == 1 and tb['img'][0].shape[0] == 2 and len(tb['img_metas'][0].data[0]) == 2
    assert tb['img_metas'][0].data[0][0]['flip'] is False


def test_voc_evaluate_hooks_into_the_fork_metric(voc):
    test = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=TEST), dict(test_mode=True))
    results = []
    for i in range(len(test)):
        a = test.get_ann_info(i)
        per = [np.zeros((0, 5), np.float32) for _ in range(20)]
        for b, l in zip(a['bboxes'], a['labels']):
            per[l] = np.vstack([per[l], np.r_[b, 0.9].astype(np.float32)[None]])
        results.append(per)
    out = test.evaluate(results, metric='mAP', logger='silent', show=False, isUnc=False, out_dir=None)
    assert out['mAP'] == 1.0 and out['AP50'] == 1.0                # perfect detections; VOC07 11-point mode (img_prefix has VOC2007)


# ------------------------------------------------------------------------------------------------------------------------------
# the same rules against the REFERENCE's own classes (tests/golden/voc_data.npz, tools/golden/make_golden_data.py: XMLDataset / VOCDataset,
# GroupSampler / DistributedGroupSampler, Resize / RandomFlip / Pad / Expand / MinIoURandomCrop / Collect run from /root/reference under the
# mmcv shim on this very tree; geometry and random-draw order only -- the mmcv pixel functions are absent)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'voc_data.npz')


def _base(h=375, w=500):
    img = np.random.RandomState(1).randint(0, 255, (h, w, 3)).astype(np.float32)
    boxes = np.array([[47, 239, 194, 370], [7, 11, 351, 297], [480, 5, 499, 60]], np.float32)
    return dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], bbox_fields=['gt_bboxes'], gt_bboxes=boxes,
                gt_labels=np.array([11, 14, 3], np.int64), filename='x.jpg', ori_filename='x.jpg')


def test_xml_dataset_matches_the_reference_golden(voc):
    g = np.load(GOLD)
    ds = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=[]))
    assert [d['id'] for d in ds.data_infos] == g['ids'].tolist() and ds.year == int(g['year'])
    assert np.array_equal([[d['width'], d['height']] for d in ds.data_infos], g['wh']) and np.array_equal(ds.flag, g['flag'])
    for i in range(len(ds)):
        a = ds.get_ann_info(i)
        for k in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore'):
            assert np.array_equal(a[k], g[f'ann{i}_{k}']) and a[k].dtype == g[f'ann{i}_{k}'].dtype, (i, k)
        assert ds.get_cat_ids(i) == g[f'cat_ids{i}'].tolist()
    test = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=[]), dict(test_mode=True))
    assert [d['id'] for d in test.data_infos] == g['test_ids'].tolist()
    dm = build_dataset(dict(type='VOCDataset', ann_file=voc + 'ImageSets/Main/trainval.txt', img_prefix=voc, pipeline=[], min_size=120))
    assert np.array_equal(dm.get_ann_info(0)['bboxes'], g['minsize_ann0_bboxes']) and np.array_equal(dm.get_ann_info(0)['bboxes_ignore'], g['minsize_ann0_ignore'])


def test_samplers_match_the_reference_golden():
    from aod_meh_hua_amd.datasets import DistributedGroupSampler
    g = np.load(GOLD)

    class Flags:
        def __init__(self, flag):
            self.flag = np.asarray(flag, dtype=np.uint8)

        def __len__(self):
            return len(self.flag)
    ds = Flags(g['sampler_flags'])
    for spg in (2, 4):
        np.random.seed(11)
        assert list(GroupSampler(ds, samples_per_gpu=spg)) == g[f'group_sampler_spg{spg}'].tolist()
    for rank in (0, 1):
        for epoch in (0, 3):
            s = DistributedGroupSampler(ds, samples_per_gpu=2, num_replicas=2, rank=rank, seed=5)
            s.set_epoch(epoch)
            assert list(s) == g[f'dist_sampler_r{rank}_e{epoch}'].tolist(), (rank, epoch)
    assert len(DistributedGroupSampler(ds, samples_per_gpu=2, num_replicas=2, rank=0, seed=5)) == int(g['dist_sampler_len'])


def test_transform_geometry_matches_the_reference_golden():
    g = np.load(GOLD)
    r = P.Resize(img_scale=(1000, 600), keep_ratio=True)(_base())
    assert tuple(r['img_shape']) == tuple(g['resize_keep_shape']) and np.array_equal(r['scale_factor'], g['resize_keep_sf']) and np.array_equal(r['gt_bboxes'], g['resize_keep_boxes'])
    r = P.Resize(img_scale=(300, 300), keep_ratio=False)(_base())
    assert tuple(r['img_shape']) == tuple(g['resize_fix_shape']) and np.array_equal(r['scale_factor'], g['resize_fix_sf']) and np.array_equal(r['gt_bboxes'], g['resize_fix_boxes'])
    r = P.Resize(img_scale=(1000, 600), keep_ratio=True)(_base(500, 333))
    assert tuple(r['img_shape']) == tuple(g['resize_tall_shape']) and np.array_equal(r['scale_factor'], g['resize_tall_sf'])
    np.random.seed(21)
    rs, sc = P.Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='range', keep_ratio=True), []
    for _ in range(8):
        d = {}
        rs._random_scale(d)
        sc.append(d['scale'])
    assert np.array_equal(np.array(sc), g['resize_range_scales'])
    np.random.seed(22)
    rs, sc = P.Resize(img_scale=[(1333, 640), (1333, 672), (1333, 800)], multiscale_mode='value', keep_ratio=True), []
    for _ in range(8):
        d = {}
        rs._random_scale(d)
        sc.append(list(d['scale']) + [d['scale_idx']])
    assert np.array_equal(np.array(sc), g['resize_value_scales'])
    code = {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}
    np.random.seed(23)
    fl = P.RandomFlip(flip_ratio=0.5)
    dec = []
    for _ in range(12):
        r = fl(_base())
        dec.append([int(bool(r['flip'])), code[r['flip_direction']]])
    assert np.array_equal(np.array(dec), g['flip_decisions'])
    np.random.seed(24)
    fl3 = P.RandomFlip(flip_ratio=[0.3, 0.2, 0.2], direction=['horizontal', 'vertical', 'diagonal'])
    dec, fb = [], []
    for _ in range(12):
        r = fl3(_base())
        dec.append(code[r['flip_direction']] if r['flip'] else 0)
        fb.append(r['gt_bboxes'])
    assert np.array_equal(np.array(dec), g['flip3_decisions']) and np.array_equal(np.stack(fb), g['flip3_boxes'])
    for d_ in ('horizontal', 'vertical', 'diagonal'):
        assert np.array_equal(P.bbox_flip(_base()['gt_bboxes'], (375, 500, 3), d_), g[f'bbox_flip_{d_}'])
    r = P.Pad(size_divisor=32)(dict(img=np.zeros((600, 800, 3), np.float32), img_fields=['img']))
    assert tuple(r['pad_shape']) == tuple(g['pad_shape'])
    np.random.seed(25)
    ex = P.Expand(mean=(123.675, 116.28, 103.53), to_rgb=True, ratio_range=(1, 4))
    shp, bx = [], []
    for _ in range(10):
        r = ex(_base(300, 400))
        shp.append(r['img'].shape[:2])
        bx.append(r['gt_bboxes'])
    assert np.array_equal(np.array(shp), g['expand_shapes']) and np.array_equal(np.stack(bx), g['expand_boxes'])
    np.random.seed(26)
    mc = P.MinIoURandomCrop(min_ious=(0.1, 0.3, 0.5, 0.7, 0.9), min_crop_size=0.3)
    for k in range(12):
        r = mc(_base(300, 400))
        assert tuple(r['img'].shape[:2]) == tuple(g['crop_shapes'][k]), k
        assert np.array_equal(r['gt_bboxes'], g[f'crop{k}_boxes']) and np.array_equal(r['gt_labels'], g[f'crop{k}_labels']), k
    np.random.seed(27)
    res = _base()
    for tr in (P.Resize(img_scale=(1000, 600), keep_ratio=True), P.RandomFlip(flip_ratio=0.5), P.Normalize(**IMG_NORM), P.Pad(size_divisor=32)):
        res = tr(res)
    meta = P.Collect(keys=['img', 'gt_bboxes', 'gt_labels'])(res)['img_metas'].data
    assert sorted(meta.keys()) == g['meta_keys'].tolist()
    assert tuple(meta['img_shape']) == tuple(g['meta_img_shape']) and tuple(meta['pad_shape']) == tuple(g['meta_pad_shape']) and tuple(meta['ori_shape']) == tuple(g['meta_ori_shape'])
    assert np.array_equal(meta['scale_factor'], g['meta_scale_factor']) and bool(meta['flip']) == bool(g['meta_flip']) and np.array_equal(res['gt_bboxes'], g['meta_boxes'])
