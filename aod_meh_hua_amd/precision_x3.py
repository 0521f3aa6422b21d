"""Debug precision mode `bf16x3` of the SAME convolution kernels (AOD_CONV_PREC=bf16x3 or functional.set_precision('bf16x3')).

The product path multiplies bf16 x bf16 into fp32 accumulators, the reference is fp32 end to end, and the parity tests therefore
carry tolerances of a few percent.  This mode shows that those residuals are operand ROUNDING and not logic: every operand is split into a
bf16 head and a bf16 tail, x = xh + xl, w = wh + wl, and

        conv(x, w) ~= conv(xh, wh) + conv(xl, wh) + conv(xh, wl)            (the dropped xl * wl term is ~2^-16 relative)

is evaluated by ONE launch of the unmodified implicit-GEMM kernel on channel-concatenated operands: activations [xh | xl | xh] (3C channels)
against weights [wh | wh | wl] -- the MFMA contraction over K = R*S*3C adds the three products in its fp32 accumulator.  dgrad uses the same
trick on the gradient channels, wgrad two launches ([xh | xl] (x) dzh and xh (x) dzl).  Activations between layers are kept in fp32 and
BN / bias / residual / ReLU and their backward run as plain torch fp32 ops: this is a measuring instrument (3x the MFMA work, ~10x the
memory traffic), not a product path -- the HIP conv / dgrad / wgrad kernels are the ones under test, none of the fused epilogues are used.

With it the golden `train_step` deviations fall from ~1e-2 to ~1e-5 (tests/test_gpu_precision_x3.py)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import hipops as ho
from .hipops import Seg


def _rows(x):
    """[B,C,H,W] fp32 (any strides) -> contiguous [B*H*W, C] fp32"""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).float().contiguous()


def _nchw(rows, B, H, W):
    return rows.view(B, H, W, rows.shape[1]).permute(0, 3, 1, 2)


def _pad8(c):
    return (c + 7) // 8 * 8


def split_hi_lo(t, cpad):
    """fp32 [M, C] -> (hi, lo) bf16 [M, cpad] with t ~= hi + lo to ~2^-16 relative"""
    hi = t.bfloat16()
    lo = (t - hi.float()).bfloat16()
    if cpad != t.shape[1]:
        hi, lo = F.pad(hi, (0, cpad - t.shape[1])), F.pad(lo, (0, cpad - t.shape[1]))
    return hi.contiguous(), lo.contiguous()


def _w_split(w):
    wh = w.detach().bfloat16().float()
    return wh, w.detach().float() - wh


class ConvFnX3(Function):
    """y = act(conv(x, w) * scale + shift + res) with ~fp32 products on the bf16 MFMA kernels (see module docstring).
    forward(ctx, meta, w, gamma, beta, mean, var, bias, res, *xs) like functional.ConvFn; tensors are fp32."""

    @staticmethod
    def forward(ctx, meta, w, gamma, beta, mean, var, bias, res, *xs):
        O, I, R, S = w.shape
        cpad = _pad8(I)
        rows = [_rows(x) for x in xs]
        x_all = torch.cat(rows) if len(rows) > 1 else rows[0]
        segs, r0 = [], 0
        for x in xs:
            segs.append(Seg(x.shape[0], x.shape[2], x.shape[3], r0))
            r0 += x.shape[0] * x.shape[2] * x.shape[3]
        xh, xl = split_hi_lo(x_all, cpad)
        wh, wl = _w_split(w)
        whp, wlp = ho.pack_weight_fwd(wh.contiguous(), cpad), ho.pack_weight_fwd(wl.contiguous(), cpad)
        w3 = torch.cat([whp, whp, wlp], dim=3).contiguous()                       # [O][R][S][3*cpad]: pairs with [xh | xl | xh]
        x3 = torch.cat([xh, xl, xh], dim=1).contiguous()
        z, zsegs = ho.conv2d_rows(x3, segs, w3, O, R, S, meta['stride'], meta['pad'], meta['dil'], out_f32=True, alg=(I, O))
        scale = invstd = None
        if gamma is not None:
            invstd = torch.rsqrt(var + meta['eps'])
            scale = gamma.detach() * invstd
            y = z * scale + (beta.detach() - mean * scale)
        elif bias is not None:
            y = z + bias.detach()
        else:
            y = z
        if res is not None:
            y = y + _rows(res)
        if meta['relu']:
            y = torch.relu(y)
        ctx.meta, ctx.segs, ctx.zsegs, ctx.nx = meta, segs, zsegs, len(xs)
        ctx.has_bn, ctx.has_bias, ctx.has_res = gamma is not None, bias is not None, res is not None
        ctx.save_for_backward(w, xh, xl, z if gamma is not None else None, y if meta['relu'] else None, scale, mean, invstd)
        return tuple(_nchw(y[s.row0:s.row0 + s.rows], s.B, s.H, s.W) for s in zsegs)

    @staticmethod
    def backward(ctx, *gouts):
        w, xh, xl, z, y, scale, mean, invstd = ctx.saved_tensors
        meta = ctx.meta
        O, I, R, S = w.shape
        cpad, opad = _pad8(I), _pad8(O)
        parts = [(_rows(g) if g is not None else torch.zeros(s.rows, O, device=w.device)) for g, s in zip(gouts, ctx.zsegs)]
        g = torch.cat(parts) if len(parts) > 1 else parts[0]
        if y is not None:
            g = g * (y > 0)
        gw = ggamma = gbeta = gbias = gres = None
        if ctx.has_res and ctx.needs_input_grad[7]:
            s = ctx.zsegs[0]
            gres = _nchw(g, s.B, s.H, s.W)
        if ctx.has_bn:
            if ctx.needs_input_grad[2]:
                ggamma = (g * ((z - mean) * invstd)).sum(0)
                gbeta = g.sum(0)
            dz = g * scale
        else:
            if ctx.has_bias and ctx.needs_input_grad[6]:
                gbias = g.sum(0)
            dz = g
        dzh, dzl = split_hi_lo(dz, opad)
        if ctx.needs_input_grad[1]:
            x2 = torch.cat([xh, xl], dim=1).contiguous()
            dw1 = torch.zeros(opad, R, S, 2 * cpad, dtype=torch.float32, device=w.device)
            ho.conv2d_wgrad_rows(x2, ctx.segs, dzh, ctx.zsegs, R, S, meta['stride'], meta['pad'], meta['dil'], dw=dw1, alg=(I, O))
            dw2 = torch.zeros(opad, R, S, cpad, dtype=torch.float32, device=w.device)
            ho.conv2d_wgrad_rows(xh, ctx.segs, dzl, ctx.zsegs, R, S, meta['stride'], meta['pad'], meta['dil'], dw=dw2, alg=(I, O))
            dw = dw1[..., :cpad] + dw1[..., cpad:] + dw2                              # xh.dzh + xl.dzh + xh.dzl
            gw = dw[:O, :, :, :I].permute(0, 3, 1, 2).contiguous()
        gxs = [None] * ctx.nx
        if any(ctx.needs_input_grad[8:]):
            wh, wl = _w_split(w)
            whd, wld = ho.pack_weight_dgrad(wh.contiguous(), opad), ho.pack_weight_dgrad(wl.contiguous(), opad)
            wd3 = torch.cat([whd, whd, wld], dim=3).contiguous()                      # [I][R][S][3*opad]: pairs with [dzh | dzl | dzh]
            dz3 = torch.cat([dzh, dzl, dzh], dim=1).contiguous()
            dx = ho.conv2d_dgrad_rows(dz3, ctx.zsegs, ctx.segs, wd3, I, R, S, meta['stride'], meta['pad'], meta['dil'], out_f32=True, alg=(I, O))
            gxs = [(_nchw(dx[s.row0:s.row0 + s.rows], s.B, s.H, s.W) if ctx.needs_input_grad[8 + i] else None) for i, s in enumerate(ctx.segs)]
        return (None, gw, ggamma, gbeta, None, None, gbias, gres) + tuple(gxs)


def conv_bn_act_x3(xl, w, bn, bias, res, meta):
    if bn is not None:
        return ConvFnX3.apply(meta, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, res, *xl)
    return ConvFnX3.apply(meta, w, None, None, None, None, bias, res, *xl)


def image_to_nhwc_x3(img):
    return img.detach().float().contiguous(memory_format=torch.channels_last)


def max_pool_x3(x):
    return F.max_pool2d(x, 3, 2, 1)


def upsample_add_x3(lateral, top):
    return lateral + F.interpolate(top, size=lateral.shape[2:], mode='nearest')
