"""FPN neck plugin (interface of mmdet/models/necks/fpn.py:10-202): 1x1 laterals, nearest top-down add,
3x3 output convs, extra stride-2 3x3 levels ('on_input' / 'on_lateral' / 'on_output'), no norm / act
(norm_cfg=None, act_cfg=None in Config_RetinaNet.py).  state_dict keys: lateral_convs.{i}.conv.*, fpn_convs.{i}.conv.*.
Every conv is one HIP implicit-GEMM launch (bias fused); the top-down add is one fused upsample+add launch."""
import torch.nn as nn

from ... import functional as AF
from ...mmcv_lite import BaseModule, ConvModule
from ..builder import NECKS


@NECKS.register_module()
class FPN(BaseModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__(init_cfg)
        assert isinstance(in_channels, list) and norm_cfg is None and act_cfg is None
        assert upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in upsample_cfg
        self.in_channels, self.out_channels, self.num_ins, self.num_outs = in_channels, out_channels, len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        assert not relu_before_extra_convs, 'relu_before_extra_convs is not used by the AL configs'
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            add_extra_convs = 'on_input' if extra_convs_on_inputs else 'on_output'
        self.add_extra_convs = add_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act_cfg=None, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act_cfg=None, inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == 'on_input') else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, act_cfg=None, inplace=False))

    def forward(self, inputs):
        """fpn.py:151-202."""
        assert len(inputs) == len(self.in_channels)
        # (shared_input: a backbone stage output also feeds the next stage -- and C5 the first extra conv --: functional.GradAcc)
        laterals = [lc(inputs[i + self.start_level], shared_input=True) for i, lc in enumerate(self.lateral_convs)]
        n = len(laterals)
        for i in range(n - 1, 0, -1):
            # (a merged lateral feeds its own output conv AND the level below: AF.fork -- a no-op handle in the bf16 mode)
            laterals[i], top = AF.fork(laterals[i], 2)
            laterals[i - 1] = AF.upsample_add(laterals[i - 1], top)
        # all output levels live in ONE pyramid buffer so that the head's level-batched convs read them in place
        from ...hipops import out_hw
        shapes = [(l.shape[0], l.shape[2], l.shape[3]) for l in laterals]
        if self.num_outs > n and self.add_extra_convs:
            src0 = inputs[self.backbone_end_level - 1] if self.add_extra_convs == 'on_input' else laterals[-1]
            hw = (src0.shape[2], src0.shape[3]) if self.add_extra_convs != 'on_output' else shapes[-1][1:]
            for _ in range(n, self.num_outs):
                hw = out_hw(hw[0], hw[1], 3, 3, 2, 1, 1)
                shapes.append((laterals[0].shape[0], hw[0], hw[1]))
        _, slots = AF.pyramid_buffer(shapes, self.out_channels, laterals[0].device)
        outs = [self.fpn_convs[i](laterals[i], out=slots[i]) for i in range(n)]
        if self.num_outs > len(outs):
            assert self.add_extra_convs, 'max-pool extra levels (Faster R-CNN style) are not on the MEH/HUA path'
            if self.add_extra_convs == 'on_input':
                src = inputs[self.backbone_end_level - 1]
            elif self.add_extra_convs == 'on_lateral':
                src = laterals[-1]
            else:
                src = outs[-1]
            outs.append(self.fpn_convs[n](src, out=slots[n], shared_input=self.add_extra_convs == 'on_input'))
            for i in range(n + 1, self.num_outs):
                outs[-1], src = AF.fork(outs[-1], 2)              # (an extra level is an output of the neck and the next extra conv's input)
                outs.append(self.fpn_convs[i](src, out=slots[i]))
        return tuple(outs)
