"""Stand-in for the third-party packages the reference imports but this image lacks.

PURPOSE: golden-vector generation ONLY (tools/golden/make_golden.py).  With this
shim installed, `/root/reference`'s RetinaNet+MEH+HUA and SSD+MEH+HUA python
paths import and run on CPU so that their outputs can be captured as fixtures
under tests/golden/.  Nothing under aod_meh_hua_amd/, tests/, bench.py or
__graft_entry__.py imports this file; /root/reference does not exist on the GPU
box.

What is real here (everything else is an inert dummy):
  mmcv.utils.Registry / build_from_cfg, mmcv.cnn.{ConvModule, build_conv_layer,
  build_norm_layer, VGG}, mmcv.runner.{BaseModule, force_fp32, auto_fp16},
  mmcv.ops.{sigmoid_focal_loss, nms, batched_nms}.

mmcv.ops.* are restatements of mmcv-full 1.3.8's published CPU semantics (the
package is not vendored in the reference; see DESIGN.md "parity unpinned at the
mmcv boundary"):
  sigmoid_focal_loss: mmcv/ops/csrc/pytorch/cuda/sigmoid_focal_loss_cuda_kernel.cuh
  nms / batched_nms:  mmcv/ops/nms.py + mmcv/ops/csrc/pytorch/nms.cpp (nms_cpu)
"""
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np
import torch
import torch.nn as nn

STUB_ROOTS = ('mmcv', 'cv2', 'torchvision', 'pycocotools', 'terminaltables',
              'wandb', 'lvis', 'cityscapesscripts', 'imagecorruptions',
              'albumentations', 'onnx', 'onnxruntime', 'seaborn', 'matplotlib',
              'sklearn')


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Dummy()

    def __iter__(self):
        return iter(())


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent
        self.children = {}
        self.build_func = build_func

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, cfg, *args, **kwargs):
        if self.build_func is not None:
            return self.build_func(cfg, *args, **kwargs, registry=self)
        return build_from_cfg(cfg, self, *args, **kwargs)

    def _register_module(self, module_class, module_name=None, force=False):
        if module_name is None:
            module_name = module_class.__name__
        if isinstance(module_name, str):
            module_name = [module_name]
        for n in module_name:
            self._module_dict[n] = module_class

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _reg(cls):
            self._register_module(cls, name, force)
            return cls
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} is not in the {registry._name} registry')
    return cls(**args)


_REGISTRY_NAMES = ('CONV_LAYERS', 'PLUGIN_LAYERS', 'ACTIVATION_LAYERS', 'NORM_LAYERS',
                   'UPSAMPLE_LAYERS', 'PADDING_LAYERS', 'RUNNERS', 'HOOKS', 'OPTIMIZERS',
                   'OPTIMIZER_BUILDERS', 'ATTENTION', 'TRANSFORMER_LAYER', 'POSITIONAL_ENCODING',
                   'TRANSFORMER_LAYER_SEQUENCE', 'FEEDFORWARD_NETWORK', 'DROPOUT_LAYERS',
                   'RUNNER_BUILDERS')


class _StubMod(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        if n in _REGISTRY_NAMES:
            v = Registry(n)
        else:
            v = type(n, (_Dummy,), {})
        setattr(self, n, v)
        return v


def _identity_decorator_factory(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]

    def deco(f):
        return f
    return deco


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, **kw):
        super().__init__()
        assert norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                              groups, bias=(bias is True or bias == 'auto'))
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def build_conv_layer(cfg, *a, **k):
    return nn.Conv2d(*a, **k)


def build_norm_layer(cfg, num_features, postfix=''):
    layer = nn.BatchNorm2d(num_features)
    for p in layer.parameters():
        p.requires_grad = cfg.get('requires_grad', True)
    return 'bn' + str(postfix), layer


def build_plugin_layer(*a, **k):
    raise NotImplementedError


class VGG(nn.Module):
    """mmcv.cnn.VGG subset used by SSDVGG (depth 16, no BN)."""
    arch = {11: (1, 1, 2, 2, 2), 13: (2, 2, 2, 2, 2), 16: (2, 2, 3, 3, 3), 19: (2, 2, 4, 4, 4)}

    def __init__(self, depth, with_bn=False, num_classes=-1, num_stages=5, dilations=(1,) * 5,
                 out_indices=(0, 1, 2, 3, 4), frozen_stages=-1, bn_eval=True, bn_frozen=False,
                 ceil_mode=False, with_last_pool=True):
        super().__init__()
        blocks = self.arch[depth][:num_stages]
        self.out_indices = out_indices
        self.inplanes = 3
        layers = []
        self.range_sub_modules = []
        start = 0
        for i, nb in enumerate(blocks):
            planes = 64 * 2 ** i if i < 4 else 512
            for _ in range(nb):
                layers.append(nn.Conv2d(self.inplanes, planes, 3, padding=dilations[i], dilation=dilations[i]))
                layers.append(nn.ReLU(inplace=True))
                self.inplanes = planes
            layers.append(nn.MaxPool2d(2, 2, ceil_mode=ceil_mode))
            end = start + nb * 2 + 1
            self.range_sub_modules.append([start, end])
            start = end
        if not with_last_pool:
            layers.pop(-1)
            self.range_sub_modules[-1][1] -= 1
        self.module_name = 'features'
        self.features = nn.Sequential(*layers)

    def init_weights(self, pretrained=None):
        pass


# ---- mmcv.ops restatements -------------------------------------------------
FLT_MIN = float(np.finfo(np.float32).tiny)


class _SFL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, gamma, alpha):
        p = torch.sigmoid(x)
        C = x.size(1)
        t = torch.nn.functional.one_hot(target.clamp(max=C), C + 1)[:, :C].to(x.dtype)
        logp = torch.log(p.clamp(min=FLT_MIN))
        log1mp = torch.log((1 - p).clamp(min=FLT_MIN))
        loss = -t * alpha * (1 - p).pow(gamma) * logp - (1 - t) * (1 - alpha) * p.pow(gamma) * log1mp
        ctx.save_for_backward(x, t)
        ctx.gamma, ctx.alpha = gamma, alpha
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        gamma, alpha = ctx.gamma, ctx.alpha
        p = torch.sigmoid(x)
        logp = torch.log(p.clamp(min=FLT_MIN))
        log1mp = torch.log((1 - p).clamp(min=FLT_MIN))
        # d/dx of the two terms (mmcv sigmoid_focal_loss_backward_cuda_kernel)
        pos = -alpha * (1 - p).pow(gamma) * (1 - p - gamma * p * logp)
        neg = (1 - alpha) * p.pow(gamma) * (gamma * (1 - p) * log1mp - p) * -1.0
        # neg term: d/dx[-(1-a) p^g log(1-p)] = -(1-a) p^g (g (1-p) log(1-p) - p)
        return g * (t * pos + (1 - t) * neg), None, None, None


def sigmoid_focal_loss(pred, target, gamma=2.0, alpha=0.25, weight=None, reduction='mean'):
    assert weight is None
    loss = _SFL.apply(pred, target, float(gamma), float(alpha))
    if reduction == 'none':
        return loss
    if reduction == 'sum':
        return loss.sum()
    return loss.sum() / pred.size(0)


def _nms_cpu(boxes, scores, iou_threshold, offset=0):
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    b = boxes.detach().cpu().float().numpy()
    s = scores.detach().cpu().float()
    order = torch.sort(s, dim=0, descending=True, stable=True)[1].numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    off = np.float32(offset)
    areas = (x2 - x1 + off) * (y2 - y1 + off)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + off)
        h = np.maximum(np.float32(0), yy2 - yy1 + off)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.as_tensor(np.array(keep, dtype=np.int64))


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    inds = _nms_cpu(boxes, scores, iou_threshold, offset).to(boxes.device)
    if max_num > 0:
        inds = inds[:max_num]
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    return dets, inds


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_cfg_.pop('type', 'nms')
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for id in torch.unique(idxs):
            mask = (idxs == id).nonzero(as_tuple=False).view(-1)
            dets, keep = nms(boxes_for_nms[mask], scores[mask], **nms_cfg_)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        keep = keep[scores[keep].argsort(descending=True)]
        boxes = boxes[keep]
        scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


def is_tuple_of(seq, expected_type):
    return isinstance(seq, tuple) and all(isinstance(i, expected_type) for i in seq)


def jit(*a, **k):
    return _identity_decorator_factory(*a, **k)


MODELS = Registry('model')

REAL = {
    'mmcv': dict(__version__='1.3.8', jit=jit, is_tuple_of=is_tuple_of,
                 mkdir_or_exist=lambda p, mode=0o777: __import__('os').makedirs(p, mode=mode, exist_ok=True)),
    'mmcv.utils': dict(Registry=Registry, build_from_cfg=build_from_cfg, is_tuple_of=is_tuple_of),
    'mmcv.cnn': dict(MODELS=MODELS, ConvModule=ConvModule, build_conv_layer=build_conv_layer,
                     build_norm_layer=build_norm_layer, build_plugin_layer=build_plugin_layer, VGG=VGG),
    'mmcv.runner': dict(BaseModule=BaseModule, force_fp32=_identity_decorator_factory,
                        auto_fp16=_identity_decorator_factory, Sequential=nn.Sequential,
                        ModuleList=nn.ModuleList),
    'mmcv.ops': dict(sigmoid_focal_loss=sigmoid_focal_loss, nms=nms, batched_nms=batched_nms),
    'mmcv.ops.nms': dict(nms=nms, batched_nms=batched_nms),
}


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubMod(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        m.__dict__.update(REAL.get(m.__name__, {}))


def install(reference_root='/root/reference'):
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


class AttrDict(dict):
    """dict with attribute access, recursive (stand-in for mmcv ConfigDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def copy(self):
        return AttrDict(dict(self))


def load_reference_model_cfg(cfg_path):
    ns = {}
    exec(open(cfg_path).read(), ns)
    cfg = AttrDict(ns['model'])
    cfg.backbone.pop('init_cfg', None)
    return cfg, ns
