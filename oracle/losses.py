"""Oracle: EDL softmax-focal, L1 / SmoothL1, MEH loss, loss parsing.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch fp32 on CPU, autograd for grads.
"""
import numpy as np
import torch

FLT_MIN = float(np.finfo(np.float32).tiny)


def sigmoid_focal_loss_none(x, target, gamma=2.0, alpha=0.25):
    """mmcv-full 1.3.8 `sigmoid_focal_loss(..., reduction='none')` (NOT in /root/reference;
    call site mmdet/models/losses/EDL_Softmax_FocalLoss.py:17).  Published kernel:
      p = sigmoid(x);  t==c: -alpha (1-p)^g log(max(p,FLT_MIN));  t!=c: -(1-alpha) p^g log(max(1-p,FLT_MIN))
    target == C (background) -> every column negative.  Cross-check: py_sigmoid_focal_loss,
    mmdet/models/losses/focal_loss.py:11-56."""
    C = x.size(1)
    p = torch.sigmoid(x)
    t = torch.nn.functional.one_hot(target.clamp(max=C), C + 1)[:, :C].to(x.dtype)
    logp = torch.log(p.clamp(min=FLT_MIN))
    log1mp = torch.log((1 - p).clamp(min=FLT_MIN))
    return -t * alpha * (1 - p).pow(gamma) * logp - (1 - t) * (1 - alpha) * p.pow(gamma) * log1mp


def edl_softmax_focal_none(cls_score, labels, gamma=2.0, alpha=0.25, loss_weight=1.0):
    """EDL_Softmax_FocalLoss.forward, reduction 'none' (EDL_Softmax_FocalLoss.py:51-69):
    prob = softmax(pred); logits = log(prob/(1-prob+1e-9)+1e-9); sigmoid focal on logits."""
    prob = cls_score.softmax(dim=1)
    eps = 1e-9
    logits = (prob / (1 - prob + eps) + eps).log()
    return loss_weight * sigmoid_focal_loss_none(logits, labels, gamma, alpha)


def loss_single(cls_score_nhwc, bbox_pred_nhwc, labels, label_weights, bbox_targets, bbox_weights,
                num_total_samples, gamma=2.0, alpha=0.25):
    """Lambda_L2Net.loss_single live branch (Lambda_L2.py:112-121) on already
    permuted inputs: cls_score [N, C], bbox_pred [N, 4] (N = B*A_l, row = (b, y, x, a)).

    loss_noR = unweighted row sum; loss_cls = sum(l * w) / avg (weight_reduce_loss,
    losses/utils.py:28-54 with reduction 'mean' + avg_factor); loss_bbox = L1
    (smooth_l1_loss.py:33-45) same reduction."""
    l = edl_softmax_focal_none(cls_score_nhwc, labels.reshape(-1), gamma, alpha)
    loss_noR = l.sum(dim=-1)
    loss_cls = (l * label_weights.reshape(-1, 1)).sum() / num_total_samples
    loss_bbox = (torch.abs(bbox_pred_nhwc - bbox_targets.reshape(-1, 4)) * bbox_weights.reshape(-1, 4)).sum() \
        / num_total_samples
    return loss_cls, loss_bbox, loss_noR


def smooth_l1(pred, target, beta=1.0):
    """mmdet/models/losses/smooth_l1_loss.py:11-28, elementwise."""
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


def meh_loss_single(L_score_flat, loss_noR, bbox_weights):
    """Lambda_L2Net.loss_single_L (Lambda_L2.py:235-241): L_score_flat [N] is the
    permuted+flattened lambda; weights = bbox_weights[..., 0]."""
    w = bbox_weights[..., 0].reshape(-1)
    return (torch.abs(L_score_flat + 1e-9 - loss_noR) * w).pow(2).mean() * 5


def parse_losses(losses):
    """SSLBase_L_Detector._parse_losses (detectors/SSL_Lambda.py:126-154): every key
    containing 'loss' is summed; list entries contribute sum(mean(each))."""
    log_vars = {}
    for k, v in losses.items():
        if torch.is_tensor(v):
            log_vars[k] = v.mean()
        else:
            s = 0.0
            for t in v:
                if torch.is_tensor(t):
                    s = s + t.mean()
            log_vars[k] = s
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    return loss, log_vars
