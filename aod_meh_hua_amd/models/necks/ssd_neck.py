"""SSDNeck + L2Norm plugins (mmdet/models/necks/ssd_neck.py:10-128): L2Norm on the first input, extra levels =
[1x1 conv + ReLU, 3x3 conv(stride, pad) + ReLU].  state_dict keys: l2_norm.weight, extra_layers.{i}.{0,1}.conv.*"""
import torch
import torch.nn as nn

from ...functional_ssd import l2norm
from ...mmcv_lite import BaseModule, ConvModule
from ..builder import NECKS


class L2Norm(nn.Module):
    def __init__(self, n_dims, scale=20., eps=1e-10):
        super().__init__()
        self.n_dims, self.eps, self.scale = n_dims, eps, scale
        self.weight = nn.Parameter(torch.full((n_dims,), float(scale)))

    def forward(self, x):
        return l2norm(x, self.weight, self.eps)


@NECKS.register_module()
class SSDNeck(BaseModule):
    def __init__(self, in_channels, out_channels, level_strides, level_paddings, l2_norm_scale=20., last_kernel_size=3, use_depthwise=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 init_cfg=[dict(type='Xavier', distribution='uniform', layer='Conv2d')]):
        super().__init__(init_cfg)
        assert not use_depthwise and norm_cfg is None
        assert len(out_channels) > len(in_channels) and len(out_channels) - len(in_channels) == len(level_strides) == len(level_paddings)
        assert tuple(in_channels) == tuple(out_channels[:len(in_channels)])
        if l2_norm_scale:
            self.l2_norm = L2Norm(in_channels[0], l2_norm_scale)
        self.extra_layers = nn.ModuleList()
        extra = out_channels[len(in_channels):]
        for i, (oc, stride, padding) in enumerate(zip(extra, level_strides, level_paddings)):
            k = last_kernel_size if i == len(extra) - 1 else 3
            self.extra_layers.append(nn.Sequential(
                ConvModule(out_channels[len(in_channels) - 1 + i], oc // 2, 1, act_cfg=act_cfg),
                ConvModule(oc // 2, oc, k, stride=stride, padding=padding, act_cfg=act_cfg)))

    def init_weights(self):
        super().init_weights()
        if hasattr(self, 'l2_norm'):
            nn.init.constant_(self.l2_norm.weight, self.l2_norm.scale)

    def forward(self, inputs):
        outs = [feat for feat in inputs]
        if hasattr(self, 'l2_norm'):
            outs[0] = self.l2_norm(outs[0])
        from ... import functional as AF
        # (every level is an output of the neck AND the input of the next extra layer: functional.fork)
        outs[-1], feat = AF.fork(outs[-1], 2) if len(self.extra_layers) else (outs[-1], None)
        for li, layer in enumerate(self.extra_layers):
            feat = layer[1](layer[0](feat))
            if li + 1 < len(self.extra_layers):
                o, feat = AF.fork(feat, 2)
                outs.append(o)
            else:
                outs.append(feat)
        return tuple(outs)
