"""EDL_Softmax_FocalLoss plugin (mmdet/models/losses/EDL_Softmax_FocalLoss.py:30-69): softmax -> logit(p) ->
sigmoid focal loss (mmcv.ops.sigmoid_focal_loss in the reference).  The module keeps the reference's
constructor and call signature; the arithmetic is the fused HIP kernel aod_edl_focal_l1_{fwd,bwd}.
Inside Lambda_L2Net.loss_single the classification and L1 box losses of one level share ONE launch
(functional.RetinaLossFn); calling this module stand-alone runs the same kernel without the box part."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import hipops as ho
from ..builder import LOSSES


class _EDLFocalFn(Function):
    @staticmethod
    def forward(ctx, pred, target, gamma, alpha):
        noR_unused, sums = None, None
        rows, C = pred.shape
        # elementwise loss [rows, C] is only needed by reduction='none' callers: recover it from the row kernel
        # by evaluating with one-hot weights is wasteful, so expose the row sum (what the hot path uses).
        loss_row, sums = ho.edl_focal_l1_fwd(pred.contiguous(), target.contiguous(), torch.ones(rows, device=pred.device), gamma=gamma, alpha=alpha)
        ctx.save_for_backward(pred, target)
        ctx.cfg = (gamma, alpha)
        return loss_row

    @staticmethod
    def backward(ctx, g_row):
        pred, target = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        zero = torch.zeros(1, device=pred.device)
        gc, _ = ho.edl_focal_l1_bwd(pred.contiguous(), target.contiguous(), torch.zeros(pred.shape[0], device=pred.device), None, None, None,
                                    zero, zero, g_row.float().contiguous(), 0.0, gamma, alpha)
        return gc, None, None, None


class _EDLFocalElemFn(Function):
    """[N, C] elementwise loss (reduction='none' / weighted reductions with per-element weights), aod_edl_focal_elem"""

    @staticmethod
    def forward(ctx, pred, target, gamma, alpha):
        pred, target = pred.contiguous(), target.contiguous()
        out = torch.empty_like(pred)
        ho.call('aod_edl_focal_elem', ho.ptr(pred), ho.ptr(target), pred.shape[0], pred.shape[1], float(gamma), float(alpha), None, ho.ptr(out), ho.stream())
        ctx.save_for_backward(pred, target)
        ctx.cfg = (gamma, alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        g = g.float().contiguous()
        out = torch.empty_like(pred)
        ho.call('aod_edl_focal_elem', ho.ptr(pred), ho.ptr(target), pred.shape[0], pred.shape[1], float(gamma), float(alpha), ho.ptr(g), ho.ptr(out), ho.stream())
        return out, None, None, None


@LOSSES.register_module()
class EDL_Softmax_FocalLoss(nn.Module):
    def __init__(self, num_classes, annealing_step, last_activation='sigmoid', gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.last_activation, self.num_classes, self.annealing_step = last_activation, num_classes, annealing_step
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """EDL_Softmax_FocalLoss.py:51-69 + the wrapper :9-27 + weight_reduce_loss (losses/utils.py:28-54).  reduction 'none' returns the
        elementwise [N, C] loss like the reference; the reduced forms with a per-row (or no) weight use the fused row kernel (the sum over
        classes commutes with a per-row weight), per-element weights go through the elementwise kernel."""
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        N, C = pred.shape
        per_elem_w = weight is not None and weight.numel() == N * C and C > 1
        if reduction == 'none' or per_elem_w:
            loss = _EDLFocalElemFn.apply(pred.float(), target, self.gamma, self.alpha)
            if weight is not None:
                loss = loss * (weight.reshape(N, -1) if weight.numel() != N else weight.reshape(N, 1))
            loss = self.loss_weight * loss
            if reduction == 'none':
                return loss
            if avg_factor is not None:
                assert reduction == 'mean', 'avg_factor can not be used with reduction="sum"'
                return loss.sum() / avg_factor
            return loss.sum() if reduction == 'sum' else loss.mean()
        row = self.loss_weight * _EDLFocalFn.apply(pred.float(), target, self.gamma, self.alpha)
        if weight is not None:
            row = row * weight.reshape(-1)
        if avg_factor is not None:
            assert reduction == 'mean', 'avg_factor can not be used with reduction="sum"'
            return row.sum() / avg_factor
        return row.sum() if reduction == 'sum' else row.sum() / (N * C)
