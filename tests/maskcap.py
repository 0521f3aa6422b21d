"""Test helper: the ReLU sign pattern of one HIP training forward, keyed like oracle.model.relu_masks expects it.

Every ReLU of the trainable network is the epilogue of a conv launch issued through functional.conv_bn_act / conv_pair_act; wrapping those
two entry points for the duration of one forward pass collects, per (layer, pyramid level), the boolean mask `output > 0` as an NCHW CPU
tensor.  (The frozen stem / layer 1 run as fused launches: no gradient passes through them, their sites keep the oracle's own ReLU.)"""
import contextlib
import re

import torch


def _key(name):
    m = re.match(r'(backbone\.layer\d+\.\d+)\.conv(\d)\.weight$', name)
    if m:
        return m.group(1) + {'1': '.bn1', '2': '.bn2', '3': '.out'}[m.group(2)], False
    m = re.match(r'(bbox_head\.(?:cls|reg|L)_convs\.\d+)\.conv\.weight$', name)
    if m:
        return m.group(1), True
    m = re.match(r'(bbox_head\.retina_L)\.weight$', name)
    if m:
        return m.group(1), True
    return None, False


@contextlib.contextmanager
def capture_relu_masks(model, masks=None):
    from aod_meh_hua_amd import functional as AF
    names = {id(p): n for n, p in model.named_parameters()}
    masks = {} if masks is None else masks
    orig_cba, orig_pair = AF.conv_bn_act, AF.conv_pair_act

    def record(w, outs):
        key, per_level = _key(names.get(id(w), ''))
        if key is None:
            return
        outs = [outs] if torch.is_tensor(outs) else list(outs)
        C = w.shape[0]
        for lvl, o in enumerate(outs):
            v = o.detach()
            v = AF.x3_to_f32(v, C) if v.dtype == torch.bfloat16 else v.float()
            masks[f'{key}@{lvl}' if per_level else key] = (v[:, :C] > 0).cpu()

    def cba(xs, w, *a, **k):
        out = orig_cba(xs, w, *a, **k)
        if k.get('relu'):
            record(w, out)
        return out

    def pair(xsA, xsB, convA, convB, *a, **k):
        out = orig_pair(xsA, xsB, convA, convB, *a, **k)
        record(convA.weight, out[0])
        record(convB.weight, out[1])
        return out

    AF.conv_bn_act, AF.conv_pair_act = cba, pair
    try:
        yield masks
    finally:
        AF.conv_bn_act, AF.conv_pair_act = orig_cba, orig_pair
