"""ResNet-50/101 backbone plugin (interface of mmdet/models/backbones/resnet.py:305-656 +
models/utils/res_layer.py:6-103): same registry name, ctor arguments used by Config_RetinaNet.py,
state_dict keys (torchvision-compatible: conv1/bn1/layerN.M.convK/bnK/downsample.{0,1}), frozen
stages and norm_eval semantics.  Compute: every conv+BN(+ReLU)(+residual) is ONE launch of the HIP
NHWC implicit-GEMM kernel with the eval-mode BN affine, the residual add and the ReLU fused in the
epilogue; activations are bf16 NHWC."""
import torch
import torch.nn as nn

from ... import functional as AF
from ...mmcv_lite import BaseModule, BatchNorm2d, Sequential, build_conv_layer, build_norm_layer
from ..builder import BACKBONES


class Bottleneck(BaseModule):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), dcn=None, plugins=None, init_cfg=None):
        super().__init__(init_cfg)
        assert style in ['pytorch', 'caffe'] and dcn is None and plugins is None
        self.inplanes, self.planes, self.stride, self.dilation, self.style = inplanes, planes, stride, dilation, style
        self.conv1_stride, self.conv2_stride = (1, stride) if style == 'pytorch' else (stride, 1)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, kernel_size=1, stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, kernel_size=3, stride=self.conv2_stride, padding=dilation,
                                      dilation=dilation, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    norm1 = property(lambda s: getattr(s, s.norm1_name))
    norm2 = property(lambda s: getattr(s, s.norm2_name))
    norm3 = property(lambda s: getattr(s, s.norm3_name))

    def forward(self, x):
        """resnet.py:262-301: relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + identity) -- 3 (4) launches."""
        # identity block: the block input feeds conv1 and, as the residual, conv3 -- nothing else (stage outputs, which also go to
        # the neck, always enter a block WITH a downsample branch), so conv1's dgrad epilogue can finish the previous block's backward
        if AF.bottleneck64_applies(self, x):       # frozen / inference 64-channel block: one launch, intermediates stay in LDS
            if AF.bottleneck64_ds_fused(self, x):      # (the first block's downsample branch rides in the same launch)
                return AF.bottleneck64_fwd(x, self, None)
            identity = x if self.downsample is None else self.downsample[0](x, bn=self.downsample[1])
            return AF.bottleneck64_fwd(x, self, identity)
        if AF.bottleneck128_applies(self, x):      # inference / frozen identity block of the 128-plane stage: one launch, filters streamed
            return AF.bottleneck128_fwd(x, self)
        # (a block WITH a downsample branch reads a stage input: conv1 and the downsample conv share it -- and with the neck's lateral conv --
        # through the gradient junction ResNet.forward put on it)
        ds = self.downsample is not None
        pre, ch = (None, None, None), None
        if AF.bottleneck128_train_applies(self, x):      # training forward of an identity block of the 128- / 256-plane stage: ONE launch
            pre = AF.bottleneck128_train_fwd(x, self)    # computes t1, t2 and y; the three calls below only record the autograd nodes around
            ch = AF.bwd_chain_for(self, x)               # them -- and their three dgrads run as one launch too (aod_bottleneck_bwd)
        role = (lambda r: (ch, r)) if ch is not None else (lambda r: None)
        out = self.conv1(x, bn=self.norm1, relu=True, sole_consumer='res' if not ds else False, shared_input=ds, pre=pre[0], chain=role(1))
        out = self.conv2(out, bn=self.norm2, relu=True, sole_consumer=True, pre=pre[1], chain=role(2))   # conv1's / conv2's outputs feed only the
        identity = x if not ds else self.downsample[0](x, bn=self.downsample[1], shared_input=True)     # next conv: their ReLU backward rides on
        return self.conv3(out, bn=self.norm3, res=identity, relu=True, sole_consumer=True, pre=pre[2], chain=role(3))       # its dgrad


class ResLayer(Sequential):
    """models/utils/res_layer.py:6-103 (downsample_first, no avg_down)."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, avg_down=False, conv_cfg=None, norm_cfg=dict(type='BN'),
                 downsample_first=True, **kwargs):
        assert not avg_down and downsample_first
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(build_conv_layer(conv_cfg, inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       build_norm_layer(norm_cfg, planes * block.expansion)[1])
        layers = [block(inplanes=inplanes, planes=planes, stride=stride, downsample=downsample, conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes=inplanes, planes=planes, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs))
        super().__init__(*layers)


@BACKBONES.register_module()
class ResNet(BaseModule):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False, frozen_stages=-1,
                 conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None, stage_with_dcn=(False,) * 4,
                 plugins=None, with_cp=False, zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert not deep_stem and not avg_down and dcn is None and plugins is None
        assert norm_eval, 'the MEH/HUA configs run BN in eval mode (norm_eval=True); train-mode BN is out of scope'
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be setting at the same time'
        if isinstance(pretrained, str):            # resnet.py:399-402 (deprecated spelling of init_cfg=Pretrained)
            self.init_cfg = dict(type='Pretrained', checkpoint=pretrained)
        elif pretrained is not None:
            raise TypeError('pretrained must be a str or None')
        if init_cfg is None and pretrained is None:
            self.init_cfg = [dict(type='Kaiming', layer='Conv2d'), dict(type='Constant', val=1, layer=['_BatchNorm', 'GroupNorm'])]
        self.zero_init_residual = zero_init_residual and init_cfg is None and pretrained is None
        self.depth, self.stem_channels = depth, stem_channels or base_channels
        self.base_channels, self.num_stages = base_channels, num_stages
        self.strides, self.dilations, self.out_indices = strides, dilations, out_indices
        assert max(out_indices) < num_stages
        self.style, self.frozen_stages, self.conv_cfg, self.norm_cfg, self.norm_eval = style, frozen_stages, conv_cfg, norm_cfg, norm_eval
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = self.stem_channels
        self.conv1 = build_conv_layer(conv_cfg, in_channels, self.stem_channels, kernel_size=7, stride=2, padding=3, bias=False)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, self.stem_channels, postfix=1)
        self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2**i
            res_layer = ResLayer(self.block, self.inplanes, planes, num_blocks, stride=strides[i], dilation=dilations[i], style=style,
                                 conv_cfg=conv_cfg, norm_cfg=norm_cfg)
            self.inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, res_layer)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * base_channels * 2**(len(self.stage_blocks) - 1)

    norm1 = property(lambda s: getattr(s, s.norm1_name))

    def init_weights(self):
        super().init_weights()
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.norm3.weight, 0)

    def _freeze_stages(self):
        """resnet.py:612-628."""
        if self.frozen_stages >= 0:
            self.norm1.eval()
            for m in [self.conv1, self.norm1]:
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        """resnet.py:630-645.  `x`: fp32 NCHW image batch (converted once to bf16 NHWC, 3 -> 8 channels)
        or an already-NHWC bf16 tensor."""
        if AF.stem_s2d_applies(x, self.conv1, self.norm1):
            x = AF.stem_pool_s2d(x, self.conv1, self.norm1)        # frozen stem: conv (space-to-depth form) + BN + ReLU + max-pool, one launch
        else:
            if x.dtype != torch.bfloat16:
                x = AF.image_to_nhwc(x, 8)
            x = self.conv1(x, bn=self.norm1, relu=True)
            x = AF.max_pool_3x3_s2(x)
        outs = []
        AF.check_junctions()
        for i, name in enumerate(self.res_layers):
            for blk in getattr(self, name):
                x = blk(x)
            x = AF.cut(x)                     # (inside functional.grad_cuts(): the backward pass of this stage becomes a segment of its own)
            AF.share_input_grad(x)            # a stage output feeds the next stage's conv1 + downsample conv and the neck (functional.GradAcc)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def grad_segments(self):
        """Parameter groups in the order their gradients become final in functional.backward_segments(): one per trainable stage, deepest
        first; everything in front of the first trainable stage's output (stem, frozen or not, and earlier stages) belongs to the last."""
        stages = [getattr(self, n) for n in self.res_layers]
        first = next((i for i, st in enumerate(stages) if any(q.requires_grad for q in st.parameters())), len(stages))
        groups = []
        for i in range(len(stages) - 1, first - 1, -1):
            mods = [stages[i]] if i > first else [self.conv1, self.norm1] + stages[:first + 1]
            groups.append([q for m in mods for q in m.parameters() if q.requires_grad])
        return groups

    def train(self, mode=True):
        """resnet.py:647-656: keep BN in eval mode, keep frozen stages frozen."""
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, BatchNorm2d):
                    m.eval()
        return self
