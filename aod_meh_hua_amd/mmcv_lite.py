"""The slice of mmcv 1.3.x that the reference's MEH/HUA path leans on, restated for this build
(mmcv is an un-vendored third-party dependency of the reference; README.md:21 pins 1.3.8).

Host plumbing only -- Registry / Config / BaseModule+init_cfg / ConvModule (parameter holder whose
compute goes to the HIP conv kernel) / data containers / a minimal BaseRunner with the hooks the
AL configs register (step LR, checkpoint, text logger, iter timer).  Interfaces follow mmcv so that
the reference's call sites read the same."""
import copy
import logging
import math
import os
import os.path as osp
import time
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import functional as AF


# ------------------------------------------------------------------------------ Registry
class Registry:
    """mmcv.utils.Registry subset: register_module(name=, module=), get (falls back to parent), build."""

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent
        self.build_func = build_func

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        return self.parent.get(key) if self.parent is not None else None

    def build(self, cfg, *args, **kwargs):
        if self.build_func is not None:
            return self.build_func(cfg, *args, **kwargs, registry=self)
        return build_from_cfg(cfg, self, *args, **kwargs)

    def _register_module(self, module_class, module_name=None, force=False):
        names = [module_class.__name__] if module_name is None else ([module_name] if isinstance(module_name, str) else module_name)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
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
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    if isinstance(t, str):
        cls = registry.get(t)
        if cls is None:
            raise KeyError(f'{t} is not in the {registry.name} registry')
    else:
        cls = t
    return cls(**args)


# ------------------------------------------------------------------------------ Config
class ConfigDict(dict):
    """dict with attribute access (mmcv ConfigDict / addict behaviour used by the reference)."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(i) for i in v]
        if isinstance(v, tuple):
            return tuple(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def copy(self):
        return ConfigDict(dict(self))

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]


class Config:
    """mmcv.Config subset: fromfile() for python configs (with `_base_` inheritance), attribute access,
    dump(), `in`, get()."""

    def __init__(self, cfg_dict=None, filename=None):
        super().__setattr__('_cfg_dict', ConfigDict(cfg_dict or {}))
        super().__setattr__('_filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = osp.abspath(osp.expanduser(filename))
        ns = {'__file__': filename}
        with open(filename) as f:
            exec(compile(f.read(), filename, 'exec'), ns)
        cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not isinstance(v, type(os)) and not callable(v)}
        base = cfg.pop('_base_', None)
        if base:
            merged = {}
            for b in ([base] if isinstance(base, str) else base):
                Config._merge(merged, Config._file2dict(osp.join(osp.dirname(filename), b)))
            Config._merge(merged, cfg)
            cfg = merged
        return cfg

    @staticmethod
    def _merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict) and not v.pop('_delete_', False):
                Config._merge(dst[k], v)
            else:
                dst[k] = v

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, k):
        return getattr(self._cfg_dict, k)

    def __setattr__(self, k, v):
        self._cfg_dict[k] = v

    def __getitem__(self, k):
        return self._cfg_dict[k]

    def __setitem__(self, k, v):
        self._cfg_dict[k] = v

    def __contains__(self, k):
        return k in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def dump(self, file=None):
        import pprint
        text = '\n'.join(f'{k} = {pprint.pformat(v)}' for k, v in self._cfg_dict.items())
        if file is None:
            return text
        with open(file, 'w') as f:
            f.write(text)

    @property
    def pretty_text(self):
        return self.dump()


def print_log(msg, logger=None, level=logging.INFO):
    """mmcv.utils.print_log: None -> print, 'silent' -> drop, Logger or logger name -> log."""
    if logger is None:
        print(msg)
    elif logger == 'silent':
        return
    elif isinstance(logger, logging.Logger):
        logger.log(level, msg)
    elif isinstance(logger, str):
        logging.getLogger(logger).log(level, msg)
    else:
        raise TypeError(f'logger should be a Logger, "silent" or None, got {type(logger)}')


def mkdir_or_exist(d, mode=0o777):
    if d:
        os.makedirs(osp.expanduser(d), mode=mode, exist_ok=True)


# ------------------------------------------------------------------------------ weight init
def _fans(w):
    rf = w[0][0].numel() if w.dim() > 2 else 1
    return w.size(1) * rf, w.size(0) * rf


def bias_init_with_prob(p):
    return float(-np.log((1 - p) / p))


def _init_one(m, cfg):
    t = cfg['type']
    w, b = getattr(m, 'weight', None), getattr(m, 'bias', None)
    bias = cfg.get('bias', 0)
    if cfg.get('bias_prob') is not None:
        bias = bias_init_with_prob(cfg['bias_prob'])
    with torch.no_grad():
        if t == 'Normal':
            if w is not None:
                w.normal_(cfg.get('mean', 0), cfg.get('std', 1))
        elif t == 'Xavier':
            if w is not None:
                fi, fo = _fans(w)
                gain = cfg.get('gain', 1)
                if cfg.get('distribution', 'normal') == 'uniform':
                    a = gain * math.sqrt(6.0 / (fi + fo))
                    w.uniform_(-a, a)
                else:
                    w.normal_(0, gain * math.sqrt(2.0 / (fi + fo)))
        elif t == 'Kaiming':
            if w is not None:
                nn.init.kaiming_normal_(w, a=cfg.get('a', 0), mode=cfg.get('mode', 'fan_out'),
                                        nonlinearity=cfg.get('nonlinearity', 'relu'))
        elif t == 'Constant':
            if w is not None:
                w.fill_(cfg['val'])
        else:
            raise KeyError(f'unknown init type {t}')
        if b is not None and isinstance(b, torch.Tensor):
            b.fill_(bias)


# mmcv resolves `torchvision://NAME` through torchvision's model_urls and `open-mmlab://NAME` through its open_mmlab.json and
# downloads into $TORCH_HOME/hub/checkpoints (mmcv/runner/checkpoint.py load_from_torchvision / load_from_openmmlab).  There is no
# network here, so the same names are resolved against local directories -- and an unresolvable one is LOUD, never a silent skip.
_PRETRAINED_ALIASES = {'vgg16_caffe': ('vgg16_caffe-292e1171.pth',), 'resnet50': ('resnet50-0676ba61.pth', 'resnet50-19c8e357.pth'),
                       'resnet101': ('resnet101-63fe2227.pth', 'resnet101-5d3b4d8f.pth')}
_pretrained_warned = set()


def pretrained_search_dirs():
    dirs = []
    if os.environ.get('AOD_PRETRAINED_DIR'):
        dirs.append(os.environ['AOD_PRETRAINED_DIR'])
    home = os.environ.get('TORCH_HOME') or osp.join(os.environ.get('XDG_CACHE_HOME', osp.expanduser('~/.cache')), 'torch')
    dirs += [osp.join(home, 'hub', 'checkpoints'), osp.join(home, 'checkpoints')]
    return dirs


def resolve_pretrained(uri):
    """Local file for a `Pretrained` checkpoint string, or None.  A plain path is returned if it exists; `torchvision://NAME`,
    `open-mmlab://NAME` and http(s) URLs are looked up by file name (the hashed names mmcv would have downloaded, `NAME.pth`, or
    `NAME-*.pth`) in $AOD_PRETRAINED_DIR, then $TORCH_HOME/hub/checkpoints (default ~/.cache/torch/hub/checkpoints)."""
    import glob
    if not uri:
        return None
    if '://' not in uri:
        return uri if osp.isfile(uri) else None
    scheme, name = uri.split('://', 1)
    if scheme in ('http', 'https'):
        names = [osp.basename(name)]
    else:
        base = osp.basename(name)
        names = list(_PRETRAINED_ALIASES.get(base, ())) + [base + '.pth', base + '-*.pth']
    for d in pretrained_search_dirs():
        for n in names:
            hits = sorted(glob.glob(osp.join(d, n)))
            if hits:
                return hits[0]
    return None


def initialize(module, init_cfg):
    """mmcv.cnn.utils.weight_init.initialize subset: {type, layer, override} + Pretrained (mmcv PretrainedInit: load_checkpoint with
    strict=False into THIS module; configs/_base_/Config_RetinaNet.py:33 `torchvision://resnet50`, Config_SSD.py:32
    `open-mmlab://vgg16_caffe`).  A checkpoint that cannot be found warns once per string and leaves the module at its constructor
    initialisation -- the synthetic benchmarks and the parity tests load seeded weights afterwards."""
    cfgs = init_cfg if isinstance(init_cfg, list) else [init_cfg]
    for cfg in cfgs:
        cfg = dict(cfg)
        if cfg['type'] == 'Pretrained':
            uri = cfg.get('checkpoint', '')
            ck = resolve_pretrained(uri)
            if ck is None:
                if uri not in _pretrained_warned:
                    _pretrained_warned.add(uri)
                    warnings.warn(f"init_cfg=dict(type='Pretrained', checkpoint={uri!r}) of {type(module).__name__}: no such file in "
                                  f"{pretrained_search_dirs()} (there is no network to download it) -- {type(module).__name__} keeps its "
                                  f"RANDOM constructor initialisation.  Put the checkpoint there (or set AOD_PRETRAINED_DIR), or remove "
                                  f"init_cfg from the config to silence this.", RuntimeWarning, stacklevel=2)
                continue
            load_checkpoint(module, ck, strict=False, prefix=cfg.get('prefix'))
            continue
        override = cfg.pop('override', None)
        layers = cfg.pop('layer', None)
        layers = [layers] if isinstance(layers, str) else layers
        if layers is not None:
            for m in module.modules():
                if type(m).__name__ in layers or any(b.__name__ in layers for b in type(m).__mro__):
                    _init_one(m, cfg)
        if override is not None:
            for ov in ([override] if isinstance(override, dict) else override):
                ov = dict(ov)
                name = ov.pop('name')
                if 'type' not in ov:
                    ov = dict(cfg, **ov)
                _init_one(getattr(module, name), ov)


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule: `init_cfg` + recursive init_weights()."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    @property
    def is_init(self):
        return self._is_init

    def init_weights(self):
        if not self._is_init:
            if self.init_cfg:
                initialize(self, self.init_cfg)
            for m in self.children():
                if hasattr(m, 'init_weights'):
                    m.init_weights()
            self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def force_fp32(*a, **k):
    def deco(f):
        return f
    return deco


auto_fp16 = force_fp32


# ------------------------------------------------------------------------------ parameter holders
class Conv2d(nn.Conv2d):
    """nn.Conv2d used as a PARAMETER HOLDER (same state_dict keys / init as the reference).  Its compute
    runs on the HIP implicit-GEMM kernel; there is no ATen fallback."""

    def forward(self, x, bn=None, res=None, relu=False, out_f32=False, out=None, sole_consumer=False, shared_input=False, pre=None, chain=None):
        assert self.groups == 1 and self.padding[0] == self.padding[1] and self.stride[0] == self.stride[1]
        return AF.conv_bn_act(x, self.weight, bn=bn, bias=self.bias, res=res, stride=self.stride[0], pad=self.padding[0],
                              dil=self.dilation[0], relu=relu, out_f32=out_f32, out=out, sole_consumer=sole_consumer, shared_input=shared_input,
                              pre=pre, chain=chain)


class BatchNorm2d(nn.BatchNorm2d):
    """Parameter/buffer holder; eval-mode statistics are folded into the producing conv's epilogue."""

    def forward(self, x):
        raise RuntimeError('BatchNorm2d is folded into the HIP conv epilogue (pass it as `bn=` to the conv)')


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv'), 'only plain Conv2d is on the MEH/HUA path'
    return Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=''):
    assert cfg['type'] == 'BN'
    layer = BatchNorm2d(num_features, eps=cfg.get('eps', 1e-5))
    for p in layer.parameters():
        p.requires_grad = cfg.get('requires_grad', True)
    return 'bn' + str(postfix), layer


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule subset (conv [+bias] [+ReLU], norm_cfg=None): attribute names `.conv`, `.activate`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kw):
        super().__init__()
        assert norm_cfg is None, 'ConvModule with norm is not on the MEH/HUA path'
        assert act_cfg is None or act_cfg['type'] == 'ReLU'
        self.conv = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                           bias=(bias is True or bias == 'auto'))
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x, out=None, sole_consumer=False, shared_input=False, pre=None):
        return self.conv(x, relu=self.with_activation, out=out, sole_consumer=sole_consumer, shared_input=shared_input, pre=pre)


# ------------------------------------------------------------------------------ data containers / parallel
class DataContainer:
    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self._data, self.stack, self.cpu_only = data, stack, cpu_only

    @property
    def data(self):
        return self._data


def scatter_kwargs(inputs, device):
    def mv(x):
        if isinstance(x, DataContainer):
            d = x.data                     # mmcv: one entry per device; this build is one process per GPU
            d = d[0] if isinstance(d, list) and len(d) == 1 else d
            return d if x.cpu_only else mv(d)
        if torch.is_tensor(x):
            return x.to(device, non_blocking=True)
        if isinstance(x, (list, tuple)):
            return type(x)(mv(i) for i in x)
        if isinstance(x, dict):
            return {k: mv(v) for k, v in x.items()}
        return x
    return mv(inputs)


class MMDataParallel(nn.Module):
    """Single-device wrapper with mmcv's `.module`, train_step/val_step scatter semantics
    (apis/train_Lambda.py:53).  Multi-GPU = one process per GPU (parallel.DistributedRunnerMixin)."""

    def __init__(self, module, device_ids=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = device_ids or [0]

    def _dev(self):
        return next(self.module.parameters()).device

    def forward(self, *inputs, **kwargs):
        return self.module(*scatter_kwargs(inputs, self._dev()), **scatter_kwargs(kwargs, self._dev()))

    def train_step(self, data, *a, **kwargs):
        return self.module.train_step(scatter_kwargs(data, self._dev()), *a, **kwargs)

    def val_step(self, data, *a, **kwargs):
        return self.module.val_step(scatter_kwargs(data, self._dev()), *a, **kwargs)


# ------------------------------------------------------------------------------ checkpoints / logging
def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, prefix=None):
    """mmcv.runner.load_checkpoint: `state_dict` wrapper and `module.` prefixes stripped, optional `prefix` sub-dict selection
    (PretrainedInit's `prefix=`); with strict=False the keys that did not land are reported (mmcv logs them), and a file none of
    whose keys match the module raises -- that is a wrong file, not a partial load."""
    ck = torch.load(filename, map_location=map_location)
    sd = ck.get('state_dict', ck) if isinstance(ck, dict) else ck
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    if prefix:
        prefix = prefix if prefix.endswith('.') else prefix + '.'
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    tgt = model.module if hasattr(model, 'module') else model
    res = tgt.load_state_dict(sd, strict=strict)
    if not strict:
        own = tgt.state_dict()
        if sd and not any(k in own for k in sd):
            raise RuntimeError(f'{filename}: none of its {len(sd)} keys (e.g. {next(iter(sd))!r}) names a tensor of '
                               f'{type(tgt).__name__} (e.g. {next(iter(own), None)!r})')
        miss = [k for k in res.missing_keys if not k.endswith('num_batches_tracked')]
        if miss or res.unexpected_keys:
            msg = (f'{osp.basename(filename)} -> {type(tgt).__name__}: {len(sd) - len(res.unexpected_keys)} tensors loaded; '
                   f'missing in file: {miss[:6]}{" ..." if len(miss) > 6 else ""}; '
                   f'unexpected in file: {res.unexpected_keys[:6]}{" ..." if len(res.unexpected_keys) > 6 else ""}')
            (logger.warning if logger is not None else print)(msg)
    return ck


def save_checkpoint(model, filename, optimizer=None, meta=None):
    tgt = model.module if hasattr(model, 'module') else model
    mkdir_or_exist(osp.dirname(filename))
    ck = {'meta': dict(meta or {}, time=time.asctime()),
          'state_dict': OrderedDict((k, v.cpu()) for k, v in tgt.state_dict().items())}
    if optimizer is not None:
        ck['optimizer'] = optimizer.state_dict()
    torch.save(ck, filename)


_loggers = {}


def get_logger(name, log_file=None, log_level=logging.INFO):
    if name in _loggers:
        return _loggers[name]
    logger = logging.getLogger(name)
    logger.propagate = False
    handlers = [logging.StreamHandler()]
    rank = int(os.environ.get('RANK', 0))
    if rank == 0 and log_file is not None:
        handlers.append(logging.FileHandler(log_file, 'w'))
    for h in handlers:
        h.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        h.setLevel(log_level)
        logger.addHandler(h)
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    _loggers[name] = logger
    return logger


class ProgressBar:
    def __init__(self, task_num=0, bar_width=50, start=True, file=None):
        self.task_num, self.completed = task_num, 0

    def update(self, n=1):
        self.completed += n


# ------------------------------------------------------------------------------ runner
class LogBuffer:
    """Accumulates log vars; accepts python floats or 0-d device tensors (synced only when averaged)."""

    def __init__(self):
        self.val_history, self.n_history, self.output, self.ready = OrderedDict(), OrderedDict(), OrderedDict(), False

    def clear(self):
        self.val_history.clear(), self.n_history.clear(), self.clear_output()

    def clear_output(self):
        self.output.clear()
        self.ready = False

    def update(self, vars, count=1):
        for k, v in vars.items():
            self.val_history.setdefault(k, []).append(v)
            self.n_history.setdefault(k, []).append(count)

    def average(self, n=0):
        for k in self.val_history:
            vals = [float(v) for v in self.val_history[k][-n:]]
            nums = np.array(self.n_history[k][-n:])
            self.output[k] = float(np.sum(np.array(vals) * nums) / np.sum(nums))
        self.ready = True


class Hook:
    def __getattr__(self, name):
        if name.startswith(('before_', 'after_')):
            return lambda runner: None
        raise AttributeError(name)

    def every_n_epochs(self, runner, n):
        return (runner.epoch + 1) % n == 0 if n > 0 else False

    def every_n_inner_iters(self, runner, n):
        return (runner.inner_iter + 1) % n == 0 if n > 0 else False


class StepLrUpdaterHook(Hook):
    """lr_config=dict(policy='step', step=[...], gamma=0.1) by epoch; touches runner.optimizer only
    (optimizer_L keeps cfg.optimizer.lr -- SURVEY 9 item 5)."""

    def __init__(self, step, gamma=0.1, by_epoch=True, warmup=None, warmup_iters=0, warmup_ratio=0.1, **kw):
        self.step, self.gamma, self.base_lr = ([step] if isinstance(step, int) else list(step)), gamma, None
        assert warmup in (None, 'linear', 'constant')
        self.warmup, self.warmup_iters, self.warmup_ratio, self.regular_lr = warmup, warmup_iters, warmup_ratio, None

    def before_run(self, runner):
        for g in runner.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in runner.optimizer.param_groups]

    def before_train_epoch(self, runner):
        exp = sum(1 for s in self.step if runner.epoch >= s)
        self.regular_lr = [lr * self.gamma ** exp for lr in self.base_lr]
        for g, lr in zip(runner.optimizer.param_groups, self.regular_lr):
            g['lr'] = lr

    def before_train_iter(self, runner):
        """mmcv LrUpdaterHook warm-up (Config_SSD.py lr_config: linear, 500 iters, ratio 0.001)."""
        if self.warmup is None or runner.iter > self.warmup_iters:
            return
        if runner.iter == self.warmup_iters:
            lrs = self.regular_lr
        elif self.warmup == 'linear':
            k = (1 - runner.iter / self.warmup_iters) * (1 - self.warmup_ratio)
            lrs = [lr * (1 - k) for lr in self.regular_lr]
        else:
            lrs = [lr * self.warmup_ratio for lr in self.regular_lr]
        for g, lr in zip(runner.optimizer.param_groups, lrs):
            g['lr'] = lr


class CheckpointHook(Hook):
    def __init__(self, interval=-1, by_epoch=True, save_optimizer=True, out_dir=None, **kw):
        self.interval, self.save_optimizer, self.out_dir = interval, save_optimizer, out_dir

    def after_train_epoch(self, runner):
        if self.every_n_epochs(runner, self.interval) and runner.rank == 0 and runner.work_dir:
            runner.save_checkpoint(self.out_dir or runner.work_dir, save_optimizer=self.save_optimizer)


class EvalHook(Hook):
    """mmdet/core/evaluation/eval_hooks.py:9-25 on mmcv's EvalHook: every `interval` epochs run single_gpu_test (isEval=True) on the
    validation loader and `dataset.evaluate(results, logger=..., **eval_kwargs)`; the metrics land in runner.log_buffer.output."""

    def __init__(self, dataloader, interval=1, by_epoch=True, start=None, save_best=None, **eval_kwargs):
        self.dataloader, self.interval, self.by_epoch, self.start = dataloader, interval, by_epoch, start
        self.eval_kwargs = eval_kwargs

    def _should_evaluate(self, runner):
        if self.start is not None and runner.epoch + 1 < self.start:
            return False
        return self.every_n_epochs(runner, self.interval)

    def after_train_epoch(self, runner):
        if not self.by_epoch or not self._should_evaluate(runner):
            return None
        from .apis.test import single_gpu_test
        results = single_gpu_test(runner.model, self.dataloader, **self.eval_kwargs)
        runner.log_buffer.output['eval_iter_num'] = len(self.dataloader)
        eval_res = self.dataloader.dataset.evaluate(results, logger=runner.logger, **self.eval_kwargs)
        for name, val in eval_res.items():
            runner.log_buffer.output[name] = val
        runner.log_buffer.ready = True
        return eval_res


class IterTimerHook(Hook):
    def before_epoch(self, runner):
        self.t = time.time()

    def before_train_epoch(self, runner):
        self.t = time.time()

    def after_train_iter(self, runner):
        runner.log_buffer.update({'time': time.time() - self.t})
        self.t = time.time()


class TextLoggerHook(Hook):
    def __init__(self, interval=100, by_epoch=True, **kw):
        self.interval = interval

    def after_train_iter(self, runner):
        if self.every_n_inner_iters(runner, self.interval):
            runner.log_buffer.average(self.interval)
            lr = runner.optimizer.param_groups[0]['lr']
            items = ', '.join(f'{k}: {v:.4f}' for k, v in runner.log_buffer.output.items())
            runner.logger.info(f'Epoch [{runner.epoch + 1}][{runner.inner_iter + 1}/{len(runner.data_loader)}]\tlr: {lr:.3e}, {items}')
            runner.log_buffer.clear_output()

    def after_train_epoch(self, runner):
        runner.log_buffer.clear()


HOOKS = Registry('hook')
for _h in (StepLrUpdaterHook, CheckpointHook, IterTimerHook, TextLoggerHook):
    HOOKS.register_module(module=_h)
RUNNERS = Registry('runner')


class BaseRunner:
    """mmcv.runner.BaseRunner subset used by MyEpochBasedRunnerLambda."""

    def __init__(self, model, batch_processor=None, optimizer=None, work_dir=None, logger=None, meta=None,
                 max_iters=None, max_epochs=None):
        self.model, self.batch_processor, self.optimizer = model, batch_processor, optimizer
        self.logger = logger or get_logger('aod')
        self.meta = meta
        self.work_dir = osp.abspath(work_dir) if work_dir else None
        mkdir_or_exist(self.work_dir)
        self.mode, self._hooks, self._epoch, self._iter, self._inner_iter = None, [], 0, 0, 0
        self._max_epochs, self._max_iters = max_epochs, max_iters
        self.log_buffer = LogBuffer()
        self.rank = int(os.environ.get('RANK', 0))
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.timestamp = None

    epoch = property(lambda s: s._epoch)
    iter = property(lambda s: s._iter)
    inner_iter = property(lambda s: s._inner_iter)
    hooks = property(lambda s: s._hooks)
    max_epochs = property(lambda s: s._max_epochs)
    max_iters = property(lambda s: s._max_iters)

    def register_hook(self, hook, priority='NORMAL'):
        self._hooks.append(hook)

    def call_hook(self, fn_name):
        out = None
        for h in self._hooks:
            r = getattr(h, fn_name)(self)
            out = r if r is not None else out
        return out

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None,
                                momentum_config=None, timer_config=dict(type='IterTimerHook')):
        if lr_config is not None:
            lc = dict(lr_config)
            policy = lc.pop('policy')
            assert policy == 'step', 'the AL configs use policy="step" only'
            self.register_hook(StepLrUpdaterHook(**lc))
        if optimizer_config is not None:
            self.register_hook(OptimizerHook(**optimizer_config))
        if checkpoint_config is not None:
            self.register_hook(CheckpointHook(**checkpoint_config))
        self.register_hook(IterTimerHook())
        if log_config is not None:
            for h in log_config.get('hooks', []):
                if h['type'] == 'TextLoggerHook':
                    self.register_hook(TextLoggerHook(interval=log_config.get('interval', 100)))

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        return load_checkpoint(self.model, filename, map_location, strict, self.logger)

    def resume(self, checkpoint, resume_optimizer=True, map_location='default'):
        ck = self.load_checkpoint(checkpoint)
        self._epoch, self._iter = ck['meta']['epoch'], ck['meta']['iter']
        if 'optimizer' in ck and resume_optimizer:
            self.optimizer.load_state_dict(ck['optimizer'])


class OptimizerHook(Hook):
    """Registered then REMOVED by train_detector_SSL (apis/train_Lambda.py:68-71): the runner steps itself."""

    def __init__(self, grad_clip=None, **kw):
        self.grad_clip = grad_clip


def build_runner(cfg, default_args=None):
    return build_from_cfg(cfg, RUNNERS, default_args)


def get_host_info():
    import getpass
    import socket
    try:
        return f'{getpass.getuser()}@{socket.gethostname()}'
    except Exception:
        return 'unknown'
