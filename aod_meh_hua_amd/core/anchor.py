"""AnchorGenerator / SSDAnchorGenerator with the reference's interface
(mmdet/core/anchor/anchor_generator.py:12-564, anchor/utils.py:4-46, anchor/builder.py).

Anchors depend only on (feature-map sizes, strides): they are computed ONCE per shape on the host with
the reference's own fp32 op order (bit-exact, tests/golden/anchors.npz) and cached on the device --
the reference recomputes them every iteration (L_anchor_head.py:144,295)."""
import numpy as np
import torch
from torch.nn.modules.utils import _pair

from ..mmcv_lite import Registry, build_from_cfg

PRIOR_GENERATORS = Registry('Generator for anchors and points')
ANCHOR_GENERATORS = PRIOR_GENERATORS


def build_prior_generator(cfg, default_args=None):
    return build_from_cfg(cfg, PRIOR_GENERATORS, default_args)


build_anchor_generator = build_prior_generator


@PRIOR_GENERATORS.register_module()
class AnchorGenerator:
    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        self.strides = [_pair(stride) for stride in strides]
        self.base_sizes = [min(stride) for stride in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None))
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2**(i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.centers, self.center_offset = scale_major, centers, center_offset
        self.base_anchors = self.gen_base_anchors()
        self._cache = {}

    @property
    def num_base_anchors(self):
        return [base_anchors.size(0) for base_anchors in self.base_anchors]

    @property
    def num_base_priors(self):
        return self.num_base_anchors

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        out = []
        for i, base_size in enumerate(self.base_sizes):
            center = self.centers[i] if self.centers is not None else None
            out.append(self.gen_single_level_base_anchors(base_size, scales=self.scales, ratios=self.ratios, center=center))
        return out

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        if center is None:
            x_center, y_center = self.center_offset * w, self.center_offset * h
        else:
            x_center, y_center = center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws, y_center + 0.5 * hs], dim=-1)

    def _meshgrid(self, x, y, row_major=True):
        xx = x.repeat(y.shape[0])
        yy = y.view(-1, 1).repeat(1, x.shape[0]).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device='cpu'):
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0, feat_w) * stride[0]
        shift_y = torch.arange(0, feat_h) * stride[1]
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        shifts = torch.stack([shift_xx, shift_yy, shift_xx, shift_yy], dim=-1).type_as(base_anchors)
        return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        """list[Tensor [H*W*A, 4]] per level, cached per (sizes, device)."""
        assert self.num_levels == len(featmap_sizes)
        key = ('a', tuple(tuple(int(v) for v in s) for s in featmap_sizes), str(device))
        if key not in self._cache:
            cpu = [self.single_level_grid_anchors(self.base_anchors[i], featmap_sizes[i], self.strides[i])
                   for i in range(self.num_levels)]
            flat = torch.cat(cpu).to(device)
            out, s = [], 0
            for a in cpu:
                out.append(flat[s:s + a.shape[0]])
                s += a.shape[0]
            self._cache[key] = (out, flat)
        return self._cache[key][0]

    grid_priors = grid_anchors

    def flat_grid_anchors(self, featmap_sizes, device='cuda'):
        self.grid_anchors(featmap_sizes, device)
        return self._cache[('a', tuple(tuple(int(v) for v in s) for s in featmap_sizes), str(device))][1]

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device='cpu'):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros(feat_w, dtype=torch.bool)
        valid_y = torch.zeros(feat_h, dtype=torch.bool)
        valid_x[:valid_w] = 1
        valid_y[:valid_h] = 1
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        valid = valid_xx & valid_yy
        return valid[:, None].expand(valid.size(0), num_base_anchors).contiguous().view(-1)

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        key = ('f', tuple(tuple(int(v) for v in s) for s in featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(device))
        if key not in self._cache:
            flags = []
            for i in range(self.num_levels):
                anchor_stride = self.strides[i]
                feat_h, feat_w = featmap_sizes[i]
                h, w = pad_shape[:2]
                valid_feat_h = min(int(np.ceil(h / anchor_stride[1])), feat_h)
                valid_feat_w = min(int(np.ceil(w / anchor_stride[0])), feat_w)
                flags.append(self.single_level_valid_flags((feat_h, feat_w), (valid_feat_h, valid_feat_w), self.num_base_anchors[i]))
            flat = torch.cat(flags).to(device)
            out, s = [], 0
            for f in flags:
                out.append(flat[s:s + f.shape[0]])
                s += f.shape[0]
            self._cache[key] = (out, flat)
        return self._cache[key][0]

    def flat_valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        self.valid_flags(featmap_sizes, pad_shape, device)
        key = ('f', tuple(tuple(int(v) for v in s) for s in featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(device))
        return self._cache[key][1]


@PRIOR_GENERATORS.register_module()
class SSDAnchorGenerator(AnchorGenerator):
    """anchor_generator.py:460-564."""

    def __init__(self, strides, ratios, basesize_ratio_range, input_size=300, scale_major=True):
        assert len(strides) == len(ratios)
        self.strides = [_pair(stride) for stride in strides]
        self.input_size = input_size
        self.centers = [(stride[0] / 2., stride[1] / 2.) for stride in self.strides]
        self.basesize_ratio_range = basesize_ratio_range
        min_ratio, max_ratio = basesize_ratio_range
        min_ratio, max_ratio = int(min_ratio * 100), int(max_ratio * 100)
        step = int(np.floor(max_ratio - min_ratio) / (self.num_levels - 2))
        min_sizes, max_sizes = [], []
        for ratio in range(int(min_ratio), int(max_ratio) + 1, step):
            min_sizes.append(int(self.input_size * ratio / 100))
            max_sizes.append(int(self.input_size * (ratio + step) / 100))
        if self.input_size == 300:
            if basesize_ratio_range[0] == 0.15:
                min_sizes.insert(0, int(self.input_size * 7 / 100)); max_sizes.insert(0, int(self.input_size * 15 / 100))
            elif basesize_ratio_range[0] == 0.2:
                min_sizes.insert(0, int(self.input_size * 10 / 100)); max_sizes.insert(0, int(self.input_size * 20 / 100))
            else:
                raise ValueError('basesize_ratio_range[0] should be either 0.15 or 0.2 when input_size is 300')
        elif self.input_size == 512:
            if basesize_ratio_range[0] == 0.1:
                min_sizes.insert(0, int(self.input_size * 4 / 100)); max_sizes.insert(0, int(self.input_size * 10 / 100))
            elif basesize_ratio_range[0] == 0.15:
                min_sizes.insert(0, int(self.input_size * 7 / 100)); max_sizes.insert(0, int(self.input_size * 15 / 100))
            else:
                raise ValueError('basesize_ratio_range[0] should be either 0.1 or 0.15 when input_size is 512')
        else:
            raise ValueError(f'Only support 300 or 512 in SSDAnchorGenerator, got {self.input_size}.')
        anchor_ratios, anchor_scales = [], []
        for k in range(len(self.strides)):
            scales = [1., np.sqrt(max_sizes[k] / min_sizes[k])]
            anchor_ratio = [1.]
            for r in ratios[k]:
                anchor_ratio += [1 / r, r]
            anchor_ratios.append(torch.Tensor(anchor_ratio))
            anchor_scales.append(torch.Tensor(scales))
        self.base_sizes, self.scales, self.ratios = min_sizes, anchor_scales, anchor_ratios
        self.scale_major, self.center_offset = scale_major, 0
        self.base_anchors = self.gen_base_anchors()
        self._cache = {}

    def gen_base_anchors(self):
        out = []
        for i, base_size in enumerate(self.base_sizes):
            base_anchors = self.gen_single_level_base_anchors(base_size, scales=self.scales[i], ratios=self.ratios[i], center=self.centers[i])
            indices = list(range(len(self.ratios[i])))
            indices.insert(1, len(indices))
            out.append(torch.index_select(base_anchors, 0, torch.LongTensor(indices)))
        return out


def images_to_levels(target, num_levels):
    """anchor/utils.py:4-17."""
    target = torch.stack(target, 0) if isinstance(target, (list, tuple)) else target
    level_targets, start = [], 0
    for n in num_levels:
        level_targets.append(target[:, start:start + n])
        start += n
    return level_targets


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """anchor/utils.py:20-46."""
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags
