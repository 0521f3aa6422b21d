"""SSDVGG backbone plugin (mmdet/models/backbones/ssd_vgg.py:12-118 on top of mmcv.cnn.VGG): VGG16 conv stages with
ceil-mode 2x2 pools, no last pool, + pool5 (3x3 s1 p1), fc6 (3x3 dilation 6), fc7 (1x1); outputs the ReLU maps at feature
indices 22 (conv4_3) and 34 (fc7).  state_dict keys `features.{idx}.{weight,bias}` as in the reference.  Each conv+ReLU is one
HIP implicit-GEMM launch; the pools are the generic max-pool kernel (trainable backbone -> backward too)."""
import torch
import torch.nn as nn

from ... import functional as AF
from ...functional_ssd import max_pool
from ...mmcv_lite import BaseModule, Conv2d
from ..builder import BACKBONES


@BACKBONES.register_module()
class SSDVGG(BaseModule):
    arch_settings = {11: (1, 1, 2, 2, 2), 13: (2, 2, 2, 2, 2), 16: (2, 2, 3, 3, 3), 19: (2, 2, 4, 4, 4)}

    def __init__(self, depth, with_last_pool=False, ceil_mode=True, out_indices=(3, 4), out_feature_indices=(22, 34), pretrained=None,
                 init_cfg=None, input_size=None, l2_norm_scale=None):
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be setting at the same time'
        if init_cfg is None and isinstance(pretrained, str):     # ssd_vgg.py:84-87 (deprecated spelling of init_cfg=Pretrained)
            init_cfg = dict(type='Pretrained', checkpoint=pretrained)
        elif pretrained is not None and not isinstance(pretrained, str):
            raise TypeError('pretrained must be a str or None')
        super().__init__(init_cfg if init_cfg is not None else [dict(type='Kaiming', layer='Conv2d')])
        layers, inpl = [], 3
        for i, nb in enumerate(self.arch_settings[depth]):
            planes = 64 * 2 ** i if i < 4 else 512
            for _ in range(nb):
                layers += [Conv2d(inpl, planes, 3, padding=1), nn.ReLU(inplace=True)]
                inpl = planes
            layers.append(nn.MaxPool2d(2, 2, ceil_mode=ceil_mode))
        if not with_last_pool:
            layers.pop(-1)
        layers += [nn.MaxPool2d(kernel_size=3, stride=1, padding=1), Conv2d(512, 1024, kernel_size=3, padding=6, dilation=6),
                   nn.ReLU(inplace=True), Conv2d(1024, 1024, kernel_size=1), nn.ReLU(inplace=True)]
        self.features = nn.Sequential(*layers)
        self.out_feature_indices = out_feature_indices

    def forward(self, x):
        """ssd_vgg.py:107-118."""
        if x.dtype != torch.bfloat16:
            x = AF.image_to_nhwc(x, 8)
        outs, i, n = [], 0, len(self.features)
        while i < n:
            layer = self.features[i]
            if isinstance(layer, nn.Conv2d):
                fuse = i + 1 < n and isinstance(self.features[i + 1], nn.ReLU)
                x = layer(x, relu=fuse)
                if fuse:
                    if i in self.out_feature_indices:
                        raise RuntimeError('pre-ReLU outputs are not used by the SSD config')
                    i += 1
            elif isinstance(layer, nn.MaxPool2d):
                k = layer.kernel_size if isinstance(layer.kernel_size, int) else layer.kernel_size[0]
                s = layer.stride if isinstance(layer.stride, int) else layer.stride[0]
                p = layer.padding if isinstance(layer.padding, int) else layer.padding[0]
                x = max_pool(x, k, s, p, layer.ceil_mode)
            if i in self.out_feature_indices:
                if i + 1 < n:
                    x, tap = AF.fork(x, 2)          # (the tapped map also feeds the next layer: see functional.fork)
                    outs.append(tap)
                else:
                    outs.append(x)
            i += 1
        return outs[0] if len(outs) == 1 else tuple(outs)
