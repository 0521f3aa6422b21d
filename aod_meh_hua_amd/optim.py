"""Optimizer plumbing: FusedSGD (torch.optim.SGD semantics on the multi-tensor HIP kernel) and the mmcv-style
`build_optimizer` used by apis/train_Lambda.py:54."""
import ctypes as C

import torch

from ._C import call, stream


class FusedSGD(torch.optim.Optimizer):
    """SGD with momentum / weight decay; one kernel launch per <= 48 tensors instead of ~5 small launches per
    parameter.  The parameter list is re-read from param_groups every step (the reference edits it in place:
    RemoveParamFromOptim, apis/train_Lambda.py:97-109)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        assert dampening == 0 and not nesterov, 'the AL configs use plain momentum SGD'
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self._lr_dev = {}          # group index -> [device fp32 scalar, value]: used once device_lr() has been called (HIP-graph replay)

    def device_lr(self):
        """Keep each group's learning rate in device memory and make step() read it from there, so that a step() captured in a HIP
        graph follows the schedule.  Call before every replay (outside capture): uploads only when a value changed."""
        for gi, group in enumerate(self.param_groups):
            ent = self._lr_dev.get(gi)
            if ent is None:
                dev = group['params'][0].device
                self._lr_dev[gi] = [torch.full((1,), float(group['lr']), dtype=torch.float32, device=dev), float(group['lr'])]
            elif ent[1] != float(group['lr']):
                ent[0].fill_(float(group['lr']))
                ent[1] = float(group['lr'])

    @torch.no_grad()
    def step(self, closure=None):
        if self._lr_dev and not torch.cuda.is_current_stream_capturing():
            self.device_lr()          # eager step between graph replays: the schedule may have changed group['lr'] since the last upload
        from . import functional as _AF
        _AF._WgradQueue.flush()       # (weight gradients still queued by a backward pass: launched before anything reads them; normally empty)
        touched, keep = [], []        # keep: contiguous gradient copies must outlive the launch that reads their raw pointers
        for gi, group in enumerate(self.param_groups):
            self._gi = gi
            ps, gs, ms, ns, first = [], [], [], [], None
            for p in group['params']:
                if p.grad is None:
                    continue
                assert p.dtype == torch.float32 and p.is_contiguous()
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                st = self.state[p]
                is_first = 'momentum_buffer' not in st
                if is_first:
                    st['momentum_buffer'] = torch.empty_like(p)
                if first is None:
                    first = is_first
                if is_first != first:      # mixed fresh/old buffers: flush what we have, start a new batch
                    self._launch(ps, gs, ms, ns, group, first)
                    ps, gs, ms, ns, first = [], [], [], [], is_first
                ps.append(p.data_ptr()), gs.append(g.data_ptr()), ms.append(st['momentum_buffer'].data_ptr()), ns.append(p.numel())
                touched.append(p)
            self._launch(ps, gs, ms, ns, group, first)
        # the kernel writes through raw pointers: tell autograd / the parameter-preparation registry (functional.ParamPrep keys on
        # tensor._version) that these parameters changed, without one no-op kernel per tensor
        if touched:
            torch._C._autograd._unsafe_set_version_counter(touched, [p._version + 1 for p in touched])

    def _launch(self, ps, gs, ms, ns, group, first):
        n = len(ps)
        if n == 0:
            return
        arr = C.c_void_p * n
        ent = self._lr_dev.get(self._gi)
        from .hipops import prof_bytes
        # algorithmic bytes: p, g, m read + p, m written = 20 B per parameter (16 on the first step: no momentum read)
        prof_bytes('sgd_multi', sum(ns) * (16 if first else 20),
                   lambda: call('aod_sgd_multi', arr(*ps), arr(*gs), arr(*ms), (C.c_int64 * n)(*ns), n, float(group['lr']),
                                C.c_void_p(ent[0].data_ptr()) if ent is not None else None, float(group['momentum']),
                                float(group['weight_decay']), int(bool(first)), float(self.grad_scale), stream()))


def build_optimizer(model, cfg):
    """mmcv.runner.build_optimizer subset: dict(type='SGD', lr, momentum, weight_decay) over all trainable params."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    assert t == 'SGD', 'the AL configs use SGD'
    cfg.pop('paramwise_cfg', None)
    module = model.module if hasattr(model, 'module') else model
    return FusedSGD([p for p in module.parameters() if p.requires_grad], **cfg)
