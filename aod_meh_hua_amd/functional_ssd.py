"""Autograd wrappers for the SSD-only kernels (csrc/ssd_ops.hip): generic max-pool, L2Norm, SSD loss."""
import math

import torch
from torch.autograd import Function

from . import hipops as ho
from ._C import call, ptr, stream
from .functional import as_nchw, as_rows


def pool_out(n, k, s, p, ceil_mode):
    """torch.nn.MaxPool2d output size (ceil_mode drops a last window that would start in the right padding)."""
    o = (math.ceil if ceil_mode else math.floor)((n + 2 * p - k) / s) + 1
    if ceil_mode and (o - 1) * s >= n + p:
        o -= 1
    return int(o)


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, s, p, ceil_mode):
        B, C, H, W = x.shape
        OH, OW = pool_out(H, k, s, p, ceil_mode), pool_out(W, k, s, p, ceil_mode)
        xr = as_rows(x)
        y = torch.empty(B * OH * OW, C, device=x.device, dtype=torch.bfloat16)
        call('aod_x3_maxpool_fwd' if ho.X3 else 'aod_maxpool_fwd', ptr(xr), ptr(y), B, H, W, C, OH, OW, k, s, p, stream())
        ctx.save_for_backward(xr)
        ctx.cfg = (B, C, H, W, OH, OW, k, s, p)
        return as_nchw(y, B, OH, OW)

    @staticmethod
    def backward(ctx, g):
        (xr,) = ctx.saved_tensors
        B, C, H, W, OH, OW, k, s, p = ctx.cfg
        gx = torch.empty_like(xr)
        call('aod_x3_maxpool_bwd' if ho.X3 else 'aod_maxpool_bwd', ptr(xr), ptr(as_rows(g)), ptr(gx), B, H, W, C, OH, OW, k, s, p, stream())
        return as_nchw(gx, B, H, W), None, None, None, None


def max_pool(x, k, s, p=0, ceil_mode=False):
    return MaxPoolFn.apply(x, k, s, p, ceil_mode)


class L2NormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        B, C, H, W = x.shape
        xr = as_rows(x)
        y = torch.empty_like(xr)
        call('aod_x3_l2norm_fwd' if ho.X3 else 'aod_l2norm_fwd', ptr(xr), ptr(weight.detach().float().contiguous()), ptr(y), xr.shape[0], C, float(eps), stream())
        ctx.save_for_backward(xr, weight)
        ctx.cfg = (B, C, H, W, eps)
        return as_nchw(y, B, H, W)

    @staticmethod
    def backward(ctx, g):
        xr, weight = ctx.saved_tensors
        B, C, H, W, eps = ctx.cfg
        gx = torch.empty_like(xr)
        gw = torch.zeros(C // 2 if ho.X3 else C, device=xr.device, dtype=torch.float32)      # (x3: C is the X-layout width)
        call('aod_x3_l2norm_bwd' if ho.X3 else 'aod_l2norm_bwd', ptr(xr), ptr(weight.detach().float().contiguous()), ptr(as_rows(g)), ptr(gx), ptr(gw), xr.shape[0], C, float(eps), stream())
        return as_nchw(gx, B, H, W), gw.to(weight.dtype), None


def l2norm(x, weight, eps):
    return L2NormFn.apply(x, weight, eps)


class SSDLossFn(Function):
    """Per image: (loss_cls_sum [B], loss_bbox_sum [B], ce [B, A]) -- My_L_ssd_head.py:182-215 (not yet divided by num_total_pos)."""

    @staticmethod
    def forward(ctx, cls, bbox, labels, label_w, bbox_t, bbox_w, num_classes, neg_pos_ratio, beta):
        B, A, C1 = cls.shape
        ctx.set_materialize_grads(False)          # (an undifferentiated output arrives as None in backward, not as a zero-filled tensor)
        cls, bbox = cls.contiguous(), bbox.contiguous()
        labels, label_w, bbox_t, bbox_w = labels.contiguous(), label_w.contiguous(), bbox_t.contiguous(), bbox_w.contiguous()
        ce = torch.empty(B, A, device=cls.device)
        sums = torch.empty(B, 3, device=cls.device)
        sel = torch.empty(B, 4, device=cls.device, dtype=torch.int32)
        call('aod_ssd_loss_fwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox), ptr(bbox_t), ptr(bbox_w), B, A, C1, int(num_classes),
             int(neg_pos_ratio), float(beta), ptr(ce), ptr(sums), ptr(sel), stream())
        ctx.save_for_backward(cls, bbox, labels, label_w, bbox_t, bbox_w, ce, sel)
        ctx.cfg = (num_classes, beta)
        return sums[:, 0], sums[:, 1], ce

    @staticmethod
    def backward(ctx, g_cls, g_box, g_ce):
        cls, bbox, labels, label_w, bbox_t, bbox_w, ce, sel = ctx.saved_tensors
        num_classes, beta = ctx.cfg
        B, A, C1 = cls.shape
        gc = None if g_cls is None else g_cls.float().contiguous()
        gb = None if g_box is None else g_box.float().contiguous()
        gn = None if g_ce is None else g_ce.float().contiguous()
        grad_cls, grad_box = torch.empty_like(cls), torch.empty_like(bbox)
        call('aod_ssd_loss_bwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox), ptr(bbox_t), ptr(bbox_w), ptr(ce), ptr(sel), B, A, C1,
             int(num_classes), float(beta), ptr(gc), ptr(gb), ptr(gn), ptr(grad_cls), ptr(grad_box), stream())
        return grad_cls, grad_box, None, None, None, None, None, None, None
