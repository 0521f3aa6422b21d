"""VERDICT r5 item 3, on the CPU: what would fp16 head + tail operand pairs buy the reference-precision mode's FORWARD pass?

The HIP kernels carry every activation / filter as a bf16 head + bf16 tail (16 significant bits) and drop the tail x tail product; §10a of
docs/LAB_NOTES.md found that the gradients' distance from the fp64 oracle (3e-4 .. 5e-3) is ReLU sign flips decided by that forward rounding.
fp16 pairs carry 22 bits at the same MFMA rate -- inside fp16's RANGE.  This script emulates both operand formats in the oracle (fp64
arithmetic; operands rounded exactly like the kernels' split: v -> fp32 -> head = rn(v), tail = rn(v - head); product = xh*wh + xl*wh + xh*wl)
with an exact backward pass through the rounded forward (straight-through), and reports, against the unrounded fp64 run:
    * ReLU elements whose sign differs,
    * gradient error per trainable tensor (worst / median) of both optimizer steps,
    * forward feature error,
    * range statistics of the fp16 variant: overflowing values, share of tails that fall into fp16's subnormal range or below its floor.
Variants: bf16x2 (today), fp16x2 (plain), fp16x2s (operands pre-scaled by a power of two per tensor so that max |v| sits at 2^12; the scale
is exact and would fold into the epilogue's per-channel scale).
    python tools/dbg/fp16_pair_study.py [B H]        # CPU only; writes profiles/r06_fp16_pair_study_{B}x{H}.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as omodel          # noqa: E402   (analysis tooling: the checker, never the product)
from tests import synth                      # noqa: E402

STATS = {}


def split(v64, fmt, scale_pow2=False, tag=None):
    """(head, tail) of v as the kernels' epilogue would store them, returned in fp64"""
    v = v64.detach().float()
    dt = torch.bfloat16 if fmt == 'bf16' else torch.float16
    k = 0.0
    if scale_pow2 and fmt == 'fp16':
        m = float(v.abs().max())
        k = float(np.floor(12 - np.log2(m))) if m > 0 else 0.0
        v = v * (2.0 ** k)
    h = v.to(dt)
    t = (v - h.float()).to(dt)
    if fmt == 'fp16' and tag is not None:
        st = STATS.setdefault(tag, dict(n=0, overflow=0, tail_subnormal=0, tail_flushed=0, small_rel=0))
        fin = torch.isfinite(h.float())
        tt = (v - h.float())[fin]
        st['n'] += v.numel()
        st['overflow'] += int((~fin).sum())
        nz = tt != 0
        st['tail_subnormal'] += int(((tt.abs() < 2.0 ** -14) & nz).sum())
        st['tail_flushed'] += int(((tt.abs() < 2.0 ** -25) & nz).sum())
    hd, td = h.double(), t.double()
    if k:
        hd, td = hd * 2.0 ** -k, td * 2.0 ** -k
    return hd, td


class QF:
    """stand-in for torch.nn.functional inside oracle.model: conv2d with rounded operands, everything else untouched"""

    def __init__(self, fmt, scaled):
        self.fmt, self.scaled = fmt, scaled

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if self.fmt is None or x.shape[1] == 3:           # (the stem reads the fp32 image through its own kernel: left exact here)
            return TF.conv2d(x, w, bias, stride, padding, dilation, groups)
        xh, xl = split(x, self.fmt, self.scaled, 'activations')
        wh, wl = split(w, self.fmt, self.scaled, 'filters')
        # straight-through: the VALUE is the three-product form on rounded operands, the GRADIENT that of the exact conv
        exact = TF.conv2d(x, w, None, stride, padding, dilation, groups)
        with torch.no_grad():
            val = TF.conv2d(xh + xl, wh + wl, None, stride, padding, dilation, groups) - TF.conv2d(xl, wl, None, stride, padding, dilation, groups)
        y = exact + (val - exact).detach()
        return y if bias is None else y + bias.view(1, -1, 1, 1)


def run(sd0, img, gtb, gtl, fmt, scaled=False):
    signs = {}
    relu0 = omodel._relu

    def relu(z, key):
        signs[key] = (z.detach() > 0)
        return relu0(z, key)
    omodel.F, omodel._relu = QF(fmt, scaled), relu
    try:
        sd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.')):
                v.requires_grad_(True)
        o = omodel.train_step(sd, img.double(), gtb, gtl)
        o['loss'].backward()
        g = {k: v.grad.detach().clone() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
        for v in sd.values():
            if v.is_floating_point():
                v.grad = None
        oL = omodel.train_step_L(sd, o['feats'], o['loss_noR'], o['targets'])
        oL['loss'].backward()
        g.update({k: v.grad.detach().clone() for k, v in sd.items() if v.is_floating_point() and v.grad is not None})
        return dict(loss=float(o['loss']), lossL=float(oL['loss']), grads=g, signs=signs, feats=[f.detach() for f in o['feats']])
    finally:
        omodel.F, omodel._relu = TF, relu0


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    sd0 = omodel.seeded_state_dict()
    gtb, gtl = synth.random_gts(B, H, W, seed=24, gmin=1, gmax=3)
    img = synth.images(B, H, W)
    ref = run(sd0, img, gtb, gtl, None)
    nrelu = sum(int(m.numel()) for m in ref['signs'].values())
    out = dict(workload=f'RetinaNet-R50 + MEH, {B} x {H} x {W}, seeded weights, both optimizer steps; fp64 arithmetic, operands rounded per variant',
               relu_elements=nrelu, variants={})
    for name, fmt, scaled in (('bf16x2 (the kernels today)', 'bf16', False), ('fp16x2', 'fp16', False), ('fp16x2, power-of-two pre-scale per tensor', 'fp16', True)):
        STATS.clear()
        r = run(sd0, img, gtb, gtl, fmt, scaled)
        flips = sum(int((r['signs'][k] != ref['signs'][k]).sum()) for k in ref['signs'])
        errs = {k: float((r['grads'][k] - ref['grads'][k]).norm() / (ref['grads'][k].norm() + 1e-300)) for k in ref['grads']}
        ferr = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(r['feats'], ref['feats']))
        worst = sorted(errs, key=errs.get, reverse=True)[:3]
        v = dict(relu_sign_flips=flips, flips_per_million=round(flips / nrelu * 1e6, 3), loss_rel_err=abs(r['loss'] - ref['loss']) / abs(ref['loss']),
                 feature_max_rel_err=ferr, grad_err_worst=max(errs.values()), grad_err_median=float(np.median(list(errs.values()))),
                 worst_tensors={k: errs[k] for k in worst})
        if fmt == 'fp16':
            v['fp16_range'] = {k: dict(values=s['n'], overflow=s['overflow'], tails_in_subnormal_range=round(s['tail_subnormal'] / max(s['n'], 1), 4),
                                       tails_below_floor=round(s['tail_flushed'] / max(s['n'], 1), 4)) for k, s in STATS.items()}
        out['variants'][name] = v
        print(name, json.dumps(v)[:400], flush=True)
    path = os.path.join(ROOT, 'profiles', f'r06_fp16_pair_study_{B}x{H}.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
