"""Data pipeline of the real VOC path: the transforms the two AL configs name, with mmdet's names, arguments, result
keys and random-number draws (mmdet/datasets/pipelines/loading.py:13-80,194-260; transforms.py:26-317,319-470,566-635,637-722,
797-905,908-1015; formating.py:193-318; test_time_aug.py:9-121; compose.py).

Host side like the reference.  Differences: images are decoded with PIL (the image ships no OpenCV) and handed over in mmcv's BGR
order; `Resize` interpolates bilinearly with half-pixel centres through torch (cv2.INTER_LINEAR without its fixed-point rounding), so
resized pixels agree to rounding, not bit for bit; everything geometric (scale factors, box arithmetic, flips, pads, crops) is exact."""
import collections
import os.path as osp

import numpy as np
import torch

from .mmcv_lite import DataContainer as DC
from .mmcv_lite import Registry, build_from_cfg

PIPELINES = Registry('pipeline')


def to_tensor(data):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, collections.abc.Sequence) and not isinstance(data, str):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


@PIPELINES.register_module()
class Compose:
    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


# ---------------------------------------------------------------------------------------------------- image helpers (mmcv.image)
def imread(path, color_type='color'):
    """mmcv.imread (cv2 backend semantics): HxWx3 uint8 in BGR order."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def rescale_size(old_size, scale):
    """mmcv.image.geometric.rescale_size: old_size (w, h); scale float or (long, short) bound."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        sf = scale
    else:
        max_long, max_short = max(scale), min(scale)
        sf = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)


def _lin_coords(n_in, n_out):
    """source index pair and weight of every destination index: half-pixel centres, clamped at the borders (cv2.INTER_LINEAR / torch
    align_corners=False: fx = (dx + 0.5) * scale - 0.5, sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= n - 1 -> (n - 1, 0))"""
    x = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
    x = np.maximum(x, np.float32(0))
    x0 = np.minimum(np.floor(x).astype(np.int64), n_in - 1)
    x1 = np.minimum(x0 + 1, n_in - 1)
    return x0, x1, (x - x0.astype(np.float32)).astype(np.float32)


def imresize(img, size):
    """mmcv.imresize(img, (w, h)), bilinear.  uint8 in -> uint8 out, float in -> float out.  Plain fp32 numpy arithmetic in a fixed order
    (rows first, then columns): the same pixels in the main process and in a loader worker, whatever the thread count -- ATen's CPU
    interpolation kernels round a handful of pixels differently with 1 and with 8 threads, which moved HUA scores of the affected images
    by ~1 % between a synchronous and a worker-backed pool loader."""
    w, h = size
    src = np.ascontiguousarray(img).astype(np.float32)
    if src.ndim == 2:
        src = src[:, :, None]
    y0, y1, wy = _lin_coords(src.shape[0], h)
    x0, x1, wx = _lin_coords(src.shape[1], w)
    wy, wx = wy[:, None, None], wx[None, :, None]
    rows = src[y0] * (np.float32(1) - wy) + src[y1] * wy                     # [h, W_in, C]
    out = rows[:, x0] * (np.float32(1) - wx) + rows[:, x1] * wx              # [h, w, C]
    if img.ndim == 2:
        out = out[:, :, 0]
    if img.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)


def imrescale(img, scale):
    h, w = img.shape[:2]
    new_w, new_h = rescale_size((w, h), scale)
    return imresize(img, (new_w, new_h)), new_w / w, new_h / h


def imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    return ((img - np.float32(mean).reshape(1, 1, -1)) / np.float32(std).reshape(1, 1, -1)).astype(np.float32)


def impad(img, shape, pad_val=0):
    out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def bgr2hsv(img):
    """cv2.cvtColor(float32 BGR in [0,255], COLOR_BGR2HSV): H in [0,360), S in [0,1], V in [0,255]."""
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = img.max(-1)
    mn = img.min(-1)
    d = v - mn
    s = np.where(v > 0, d / np.maximum(v, 1e-12), 0).astype(np.float32)
    h = np.zeros_like(v)
    nz = d > 0
    idx = nz & (v == r)
    h[idx] = (60 * (g - b) / np.maximum(d, 1e-12))[idx]
    idx = nz & (v == g) & (v != r)
    h[idx] = (120 + 60 * (b - r) / np.maximum(d, 1e-12))[idx]
    idx = nz & (v == b) & (v != r) & (v != g)
    h[idx] = (240 + 60 * (r - g) / np.maximum(d, 1e-12))[idx]
    h = np.where(h < 0, h + 360, h)
    return np.stack([h, s, v], -1).astype(np.float32)


def hsv2bgr(img):
    h, s, v = img[..., 0], img[..., 1], img[..., 2]
    hi = np.floor(h / 60.0).astype(np.int32) % 6
    f = h / 60.0 - np.floor(h / 60.0)
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    r = np.choose(hi, [v, q, p, p, t, v])
    g = np.choose(hi, [t, v, v, q, p, p])
    b = np.choose(hi, [p, p, t, v, v, q])
    return np.stack([b, g, r], -1).astype(np.float32)


# ---------------------------------------------------------------------------------------------------- loading
@PIPELINES.register_module()
class LoadImageFromFile:
    def __init__(self, to_float32=False, color_type='color', file_client_args=None):
        self.to_float32, self.color_type = to_float32, color_type

    def __call__(self, results):
        if results['img_prefix'] is not None:
            filename = osp.join(results['img_prefix'], results['img_info']['filename'])
        else:
            filename = results['img_info']['filename']
        img = imread(filename, self.color_type)
        if self.to_float32:
            img = img.astype(np.float32)
        results.update(filename=filename, ori_filename=results['img_info']['filename'], img=img, img_shape=img.shape, ori_shape=img.shape,
                       img_fields=['img'])
        return results


@PIPELINES.register_module()
class LoadAnnotations:
    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True, file_client_args=None):
        assert not with_mask and not with_seg, 'masks are not on the MEH/HUA path'
        self.with_bbox, self.with_label = with_bbox, with_label

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            if ann.get('bboxes_ignore') is not None:
                results['gt_bboxes_ignore'] = ann['bboxes_ignore'].copy()
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        return results


# ---------------------------------------------------------------------------------------------------- geometric transforms
@PIPELINES.register_module()
class Resize:
    """transforms.py:26-317 (multiscale 'range' / 'value', ratio_range, keep_ratio, bbox_clip_border)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, bbox_clip_border=True, backend='cv2',
                 override=False):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
        assert multiscale_mode in ('value', 'range')
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio
        self.bbox_clip_border, self.override = bbox_clip_border, override

    @staticmethod
    def random_select(img_scales):
        idx = np.random.randint(len(img_scales))
        return img_scales[idx], idx

    @staticmethod
    def random_sample(img_scales):
        long_e = [max(s) for s in img_scales]
        short_e = [min(s) for s in img_scales]
        long_edge = np.random.randint(min(long_e), max(long_e) + 1)
        short_edge = np.random.randint(min(short_e), max(short_e) + 1)
        return (long_edge, short_edge), None

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        lo, hi = ratio_range
        ratio = np.random.random_sample() * (hi - lo) + lo
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            scale, idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            scale, idx = self.random_sample(self.img_scale)
        else:
            scale, idx = self.random_select(self.img_scale)
        results['scale'], results['scale_idx'] = scale, idx

    def _resize_img(self, results):
        for key in results.get('img_fields', ['img']):
            if self.keep_ratio:
                h, w = results[key].shape[:2]
                img, _, _ = imrescale(results[key], results['scale'])
                new_h, new_w = img.shape[:2]
                w_scale, h_scale = new_w / w, new_h / h
            else:
                h, w = results[key].shape[:2]
                img = imresize(results[key], tuple(results['scale']))
                w_scale, h_scale = results['scale'][0] / w, results['scale'][1] / h
            results[key] = img
            results['img_shape'] = img.shape
            results['pad_shape'] = img.shape
            results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
            results['keep_ratio'] = self.keep_ratio

    def _resize_bboxes(self, results):
        for key in results.get('bbox_fields', []):
            bboxes = results[key] * results['scale_factor']
            if self.bbox_clip_border:
                img_shape = results['img_shape']
                bboxes[:, 0::2] = np.clip(bboxes[:, 0::2], 0, img_shape[1])
                bboxes[:, 1::2] = np.clip(bboxes[:, 1::2], 0, img_shape[0])
            results[key] = bboxes

    def __call__(self, results):
        if 'scale' not in results:
            if 'scale_factor' in results:
                img_shape = results['img'].shape[:2]
                sf = results['scale_factor']
                assert isinstance(sf, float)
                results['scale'] = tuple([int(x * sf) for x in img_shape][::-1])
            else:
                self._random_scale(results)
        else:
            if not self.override:
                assert 'scale_factor' not in results, 'scale and scale_factor cannot be both set.'
            else:
                results.pop('scale')
                results.pop('scale_factor', None)
                self._random_scale(results)
        self._resize_img(results)
        self._resize_bboxes(results)
        return results


def bbox_flip(bboxes, img_shape, direction):
    assert bboxes.shape[-1] % 4 == 0
    flipped = bboxes.copy()
    if direction == 'horizontal':
        w = img_shape[1]
        flipped[..., 0::4] = w - bboxes[..., 2::4]
        flipped[..., 2::4] = w - bboxes[..., 0::4]
    elif direction == 'vertical':
        h = img_shape[0]
        flipped[..., 1::4] = h - bboxes[..., 3::4]
        flipped[..., 3::4] = h - bboxes[..., 1::4]
    elif direction == 'diagonal':
        w, h = img_shape[1], img_shape[0]
        flipped[..., 0::4] = w - bboxes[..., 2::4]
        flipped[..., 1::4] = h - bboxes[..., 3::4]
        flipped[..., 2::4] = w - bboxes[..., 0::4]
        flipped[..., 3::4] = h - bboxes[..., 1::4]
    else:
        raise ValueError(f"Invalid flipping direction '{direction}'")
    return flipped


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:319-470 (single ratio / single direction form used by the configs; list forms draw like the reference)."""

    def __init__(self, flip_ratio=None, direction='horizontal'):
        self.flip_ratio, self.direction = flip_ratio, direction

    def __call__(self, results):
        if 'flip' not in results:
            direction_list = self.direction if isinstance(self.direction, list) else [self.direction]
            direction_list = list(direction_list) + [None]
            if isinstance(self.flip_ratio, list):
                non_flip = 1 - sum(self.flip_ratio)
                ratio_list = self.flip_ratio + [non_flip]
            else:
                non_flip = 1 - self.flip_ratio
                single = self.flip_ratio / (len(direction_list) - 1)
                ratio_list = [single] * (len(direction_list) - 1) + [non_flip]
            cur_dir = np.random.choice(direction_list, p=ratio_list)
            results['flip'] = cur_dir is not None
        if 'flip_direction' not in results:
            results['flip_direction'] = cur_dir
        if results['flip']:
            for key in results.get('img_fields', ['img']):
                img = results[key]
                fd = results['flip_direction']
                results[key] = np.ascontiguousarray(img[:, ::-1] if fd == 'horizontal' else (img[::-1] if fd == 'vertical' else img[::-1, ::-1]))
            for key in results.get('bbox_fields', []):
                results[key] = bbox_flip(results[key], results['img_shape'], results['flip_direction'])
        return results


@PIPELINES.register_module()
class Pad:
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None)
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            img = results[key]
            if self.size is not None:
                shape = self.size
            else:
                d = self.size_divisor
                shape = (int(np.ceil(img.shape[0] / d)) * d, int(np.ceil(img.shape[1] / d)) * d)
            results[key] = impad(img, shape, self.pad_val)
        results['pad_shape'] = results['img'].shape
        results['pad_fixed_size'], results['pad_size_divisor'] = self.size, self.size_divisor
        return results


@PIPELINES.register_module()
class Normalize:
    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            results[key] = imnormalize(results[key], self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results


# ---------------------------------------------------------------------------------------------------- SSD augmentations
@PIPELINES.register_module()
class PhotoMetricDistortion:
    """transforms.py:797-905: brightness, then contrast first or last (coin), saturation, hue, channel swap; every step with p = 0.5
    (np.random.randint(2)).  Same sequence of random draws as the reference."""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta, self.hue_delta = brightness_delta, hue_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range

    def __call__(self, results):
        assert results.get('img_fields', ['img']) == ['img']
        img = results['img']
        assert img.dtype == np.float32, 'PhotoMetricDistortion needs float32 images (LoadImageFromFile(to_float32=True))'
        rnd = np.random
        if rnd.randint(2):
            img = img + rnd.uniform(-self.brightness_delta, self.brightness_delta)
        mode = rnd.randint(2)
        if mode == 1 and rnd.randint(2):
            img = img * rnd.uniform(self.contrast_lower, self.contrast_upper)
        img = bgr2hsv(img.astype(np.float32))
        if rnd.randint(2):
            img[..., 1] *= rnd.uniform(self.saturation_lower, self.saturation_upper)
        if rnd.randint(2):
            img[..., 0] += rnd.uniform(-self.hue_delta, self.hue_delta)
            img[..., 0][img[..., 0] > 360] -= 360
            img[..., 0][img[..., 0] < 0] += 360
        img = hsv2bgr(img)
        if mode == 0 and rnd.randint(2):
            img = img * rnd.uniform(self.contrast_lower, self.contrast_upper)
        if rnd.randint(2):
            img = img[..., rnd.permutation(3)]
        results['img'] = np.ascontiguousarray(img.astype(np.float32))
        return results


@PIPELINES.register_module()
class Expand:
    """transforms.py:908-990: with probability `prob` paste the image at a random offset into a mean-filled canvas `ratio` x larger."""

    def __init__(self, mean=(0, 0, 0), to_rgb=True, ratio_range=(1, 4), seg_ignore_label=None, prob=0.5):
        self.to_rgb, self.ratio_range = to_rgb, ratio_range
        self.mean = mean[::-1] if to_rgb else mean
        self.min_ratio, self.max_ratio = ratio_range
        self.prob = prob

    def __call__(self, results):
        if np.random.uniform(0, 1) > self.prob:
            return results
        img = results['img']
        h, w, c = img.shape
        ratio = np.random.uniform(self.min_ratio, self.max_ratio)
        expand_img = np.full((int(h * ratio), int(w * ratio), c), self.mean, dtype=img.dtype)
        left = int(np.random.uniform(0, w * ratio - w))
        top = int(np.random.uniform(0, h * ratio - h))
        expand_img[top:top + h, left:left + w] = img
        results['img'] = expand_img
        for key in results.get('bbox_fields', []):
            results[key] = results[key] + np.tile((left, top), 2).astype(results[key].dtype)
        return results


@PIPELINES.register_module()
class MinIoURandomCrop:
    """transforms.py:993-1104: pick a min-IoU mode, try up to 50 crops (size 0.3..1 of the image, aspect in [0.5, 2]) until every gt
    overlaps the patch by at least the mode's IoU; keep the boxes whose centre falls inside, clip and shift them."""

    def __init__(self, min_ious=(0.1, 0.3, 0.5, 0.7, 0.9), min_crop_size=0.3, bbox_clip_border=True):
        self.min_ious, self.sample_mode = min_ious, (1, *min_ious, 0)
        self.min_crop_size, self.bbox_clip_border = min_crop_size, bbox_clip_border
        self.bbox2label = {'gt_bboxes': 'gt_labels', 'gt_bboxes_ignore': 'gt_labels_ignore'}

    def __call__(self, results):
        from .core.evaluation import bbox_overlaps
        img = results['img']
        assert 'bbox_fields' in results
        boxes = np.concatenate([results[key] for key in results['bbox_fields']], 0)
        h, w, c = img.shape
        while True:
            mode = np.random.choice(self.sample_mode)
            self.mode = mode
            if mode == 1:
                return results
            min_iou = mode
            for _ in range(50):
                new_w = np.random.uniform(self.min_crop_size * w, w)
                new_h = np.random.uniform(self.min_crop_size * h, h)
                if new_h / new_w < 0.5 or new_h / new_w > 2:
                    continue
                left = np.random.uniform(w - new_w)
                top = np.random.uniform(h - new_h)
                patch = np.array((int(left), int(top), int(left + new_w), int(top + new_h)))
                if patch[2] == patch[0] or patch[3] == patch[1]:
                    continue
                overlaps = bbox_overlaps(patch.reshape(-1, 4).astype(np.float32), boxes.reshape(-1, 4).astype(np.float32)).reshape(-1)
                if len(overlaps) > 0 and overlaps.min() < min_iou:
                    continue
                if len(overlaps) > 0:
                    def is_center_of_bboxes_in_patch(bx, pt):
                        center = (bx[:, :2] + bx[:, 2:]) / 2
                        return (center[:, 0] > pt[0]) * (center[:, 1] > pt[1]) * (center[:, 0] < pt[2]) * (center[:, 1] < pt[3])
                    mask = is_center_of_bboxes_in_patch(boxes, patch)
                    if not mask.any():
                        continue
                    for key in results.get('bbox_fields', []):
                        bx = results[key].copy()
                        mask = is_center_of_bboxes_in_patch(bx, patch)
                        bx = bx[mask]
                        if self.bbox_clip_border:
                            bx[:, 2:] = bx[:, 2:].clip(max=patch[2:])
                            bx[:, :2] = bx[:, :2].clip(min=patch[:2])
                        bx -= np.tile(patch[:2], 2)
                        results[key] = bx
                        label_key = self.bbox2label.get(key)
                        if label_key in results:
                            results[label_key] = results[label_key][mask]
                results['img'] = img[patch[1]:patch[3], patch[0]:patch[2]]
                results['img_shape'] = results['img'].shape
                return results


# ---------------------------------------------------------------------------------------------------- formatting
@PIPELINES.register_module()
class ImageToTensor:
    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for key in self.keys:
            img = results[key]
            if len(img.shape) < 3:
                img = np.expand_dims(img, -1)
            results[key] = to_tensor(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return results


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:193-254."""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            results.setdefault('pad_shape', img.shape)
            results.setdefault('scale_factor', 1.0)
            num_channels = 1 if len(img.shape) < 3 else img.shape[2]
            results.setdefault('img_norm_cfg', dict(mean=np.zeros(num_channels, dtype=np.float32), std=np.ones(num_channels, dtype=np.float32),
                                                    to_rgb=False))
            if len(img.shape) < 3:
                img = np.expand_dims(img, -1)
            results['img'] = DC(to_tensor(np.ascontiguousarray(img.transpose(2, 0, 1))), stack=True)
        for key in ['proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels']:
            if key in results:
                results[key] = DC(to_tensor(results[key]))
        return results


@PIPELINES.register_module()
class Collect:
    """formating.py:257-318."""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                                        'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {}
        data['img_metas'] = DC({k: results[k] for k in self.meta_keys if k in results}, cpu_only=True)
        for key in self.keys:
            data[key] = results[key]
        return data


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:9-121: one result dict per (scale, flip) combination, values gathered into lists."""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        assert (img_scale is None) ^ (scale_factor is None)
        if img_scale is not None:
            self.img_scale, self.scale_key = (img_scale if isinstance(img_scale, list) else [img_scale]), 'scale'
        else:
            self.img_scale, self.scale_key = (scale_factor if isinstance(scale_factor, list) else [scale_factor]), 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]

    def __call__(self, results):
        aug_data = []
        flip_args = [(False, None)]
        if self.flip:
            flip_args += [(True, d) for d in self.flip_direction]
        for scale in self.img_scale:
            for flip, direction in flip_args:
                _results = results.copy()
                _results[self.scale_key] = scale
                _results['flip'], _results['flip_direction'] = flip, direction
                aug_data.append(self.transforms(_results))
        out = {key: [] for key in aug_data[0]}
        for data in aug_data:
            for key, val in data.items():
                out[key].append(val)
        return out
