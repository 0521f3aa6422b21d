"""GPU parity tests of the HIP kernels (through the C ABI) against the CPU oracle.

Conv kernels compute bf16 x bf16 -> fp32; the oracle is torch's fp32 CPU conv evaluated on the
SAME bf16-rounded operands, so the only differences are fp32 summation order and the final
bf16 rounding of the output (2^-9 relative) -- tolerances below are set for that."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses
from tests import synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("bf16_mode")]      # the FAST mode's own kernels (plain bf16 operands handed to hipops directly)

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def ho():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from aod_meh_hua_amd import hipops
    return hipops


def bf(x):
    return x.to(torch.bfloat16).float()


def nhwc_rows(x_nchw):
    B, C, H, W = x_nchw.shape
    return x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def rows_nchw(rows, B, H, W):
    return rows.view(B, H, W, -1).permute(0, 3, 1, 2)


def close(a, b, rtol, atol):
    a, b = a.float().cpu(), b.float().cpu()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        d = (a - b).abs()
        i = d.argmax()
        print('max abs diff', float(d.max()), 'at', int(i), float(a.flatten()[i]), float(b.flatten()[i]), 'ref absmax', float(b.abs().max()))
    return ok


CONV_CASES = [
    # B, C, H, W, N, R, stride, pad, dil
    (2, 64, 16, 16, 64, 1, 1, 0, 1),
    (2, 64, 16, 16, 256, 3, 1, 1, 1),
    (3, 256, 20, 12, 256, 3, 1, 1, 1),      # ragged M (720 rows), K = 2304
    (2, 128, 17, 19, 128, 3, 2, 1, 1),      # stride 2, odd sizes
    (2, 256, 16, 16, 512, 1, 2, 0, 1),      # 1x1 stride 2 (downsample)
    (1, 8, 40, 40, 64, 7, 2, 3, 1),         # stem-like (C padded to 8, K = 392: K tail)
    (2, 256, 9, 9, 180, 3, 1, 1, 1),        # retina_cls: N = 180 (N tail)
    (2, 256, 9, 9, 36, 3, 1, 1, 1),
    (2, 256, 9, 9, 9, 3, 1, 1, 1),          # retina_L: N = 9 (scalar store path)
    (1, 512, 10, 10, 1024, 3, 1, 6, 6),     # dilated (SSD fc6-like)
    (4, 2048, 4, 4, 256, 3, 2, 1, 1),       # FPN extra conv on C5, K = 18432
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward(ho, case):
    B, C, H, W, N, R, stride, pad, dil = case
    g = synth.gen(100 + N + C)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(N, C, R, R, generator=g) / np.sqrt(C * R * R))
    out_f32 = N in (180, 36, 9)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    xr = nhwc_rows(x).cuda().to(torch.bfloat16)
    wp = ho.pack_weight_fwd(w.cuda())
    y, segs = ho.conv2d_rows(xr, [ho.Seg(B, H, W)], wp, N, R, R, stride, pad, dil, out_f32=out_f32)
    torch.cuda.synchronize()
    got = rows_nchw(y, B, segs[0].H, segs[0].W)
    assert got.shape == ref.shape
    assert close(got, ref, rtol=1e-2 if not out_f32 else 1e-4, atol=1e-2 if not out_f32 else 1e-4)


def test_conv_segment_with_more_than_4m_rows(ho):
    """Row decode: segments below 2^22 pixels use a float-reciprocal division (exact there); a larger one must take the integer path."""
    B, C, H, W, N = 1, 8, 2048, 2056, 16
    g = synth.gen(5)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9))
    ref = F.conv2d(x, w, None, 1, 1, 1)
    y, segs = ho.conv2d_rows(nhwc_rows(x).cuda().bfloat16(), [ho.Seg(B, H, W)], ho.pack_weight_fwd(w.cuda()), N, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    assert close(rows_nchw(y, B, H, W), ref, 1e-2, 1e-2)


def test_conv_epilogue_bn_res_relu_and_z(ho):
    B, C, H, W, N = 2, 128, 12, 12, 256
    g = synth.gen(7)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(N, C, 1, 1, generator=g) / np.sqrt(C))
    scale = torch.rand(N, generator=g) + 0.5
    shift = torch.randn(N, generator=g)
    res = bf(torch.randn(B, N, H, W, generator=g))
    z_ref = F.conv2d(x, w)
    ref = F.relu(z_ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    y, segs, z = ho.conv2d_rows(nhwc_rows(x).cuda().bfloat16(), [ho.Seg(B, H, W)], ho.pack_weight_fwd(w.cuda()), N, 1, 1,
                                pre_scale=scale.cuda(), pre_shift=shift.cuda(), res=nhwc_rows(res).cuda().bfloat16(), relu=True, save_z=True)
    torch.cuda.synchronize()
    assert close(rows_nchw(y, B, H, W), ref, 1e-2, 2e-2)
    assert close(rows_nchw(z, B, H, W), z_ref, 1e-2, 1e-2)


def test_conv_pyramid_segments_share_weights(ho):
    """Level-batched launch: 5 pyramid levels in one flat row buffer == 5 separate convs."""
    B, C, N = 2, 256, 256
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    g = synth.gen(11)
    xs = [bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes]
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9))
    bias = torch.randn(N, generator=g)
    segs, r = [], 0
    for h, w_ in sizes:
        segs.append(ho.Seg(B, h, w_, r))
        r += B * h * w_
    rows = torch.cat([nhwc_rows(x) for x in xs]).cuda().bfloat16()
    y, osegs = ho.conv2d_rows(rows, segs, ho.pack_weight_fwd(w.cuda()), N, 3, 3, 1, 1, 1, pre_shift=bias.cuda(), relu=True)
    torch.cuda.synchronize()
    for x, s in zip(xs, osegs):
        ref = F.relu(F.conv2d(x, w, bias, 1, 1))
        assert close(ho.rows_to_nchw(y, s), ref, 1e-2, 1e-2)


@pytest.mark.parametrize('N,out_f32,relu', [(180, True, False), (36, True, False), (9, True, True), (72, False, True), (200, False, False)])
def test_halo_tile_conv_over_pyramid_levels(ho, N, out_f32, relu):
    """csrc/halo_conv.hip (prediction convs, Lambda_L2.py:52-54,92-103): five ragged pyramid levels in one launch -- tiles that straddle the
    image border, levels smaller than one 8 x 16 tile, N with a 4- / 1-channel tail -- against fp32 torch on the bf16-rounded operands, and
    against the general implicit-GEMM kernel (AOD_HALO_CONV=0 path)."""
    B, C = 3, 256
    sizes = [(19, 21), (10, 11), (8, 16), (5, 3), (1, 2)]
    g = synth.gen(300 + N)
    xs = [bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes]
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9))
    bias = torch.randn(N, generator=g)
    segs, r = [], 0
    for h, w_ in sizes:
        segs.append(ho.Seg(B, h, w_, r))
        r += B * h * w_
    rows = torch.cat([nhwc_rows(x) for x in xs]).cuda().bfloat16()
    wp = ho.pack_weight_fwd(w.cuda())
    assert ho.HALO_CONV
    keep_all, ho.HALO_ALL = ho.HALO_ALL, True            # every qualifying launch (the default routes N <= 64 forward launches only)
    try:
        y, osegs = ho.conv2d_rows(rows, segs, wp, N, 3, 3, 1, 1, 1, pre_shift=bias.cuda(), relu=relu, out_f32=out_f32)
        ho.HALO_CONV = False
        y0, _ = ho.conv2d_rows(rows, segs, wp, N, 3, 3, 1, 1, 1, pre_shift=bias.cuda(), relu=relu, out_f32=out_f32)
    finally:
        ho.HALO_CONV, ho.HALO_ALL = True, keep_all
    torch.cuda.synchronize()
    tol = 2e-4 if out_f32 else 1e-2
    for x, s_ in zip(xs, osegs):
        ref = F.conv2d(x, w, bias, 1, 1)
        ref = F.relu(ref) if relu else ref
        assert close(ho.rows_to_nchw(y, s_), ref, tol, tol)
    assert close(y, y0, tol, tol)                 # (different K order: chunk-major here, tap-major there)


@pytest.mark.parametrize('N', [180, 36, 9])
def test_halo_tile_dgrad_with_fused_relu_mask_and_bias_sums(ho, N):
    """dgrad of a prediction conv through the halo-tile kernel (mirrored taps on the dgrad packing) with the producer's ReLU mask and the
    column sums of the masked gradient fused (functional.ActSlot): vs fp32 autograd and vs the general kernel."""
    B, C = 2, 256
    sizes = [(13, 18), (7, 9), (4, 4)]
    g = synth.gen(330 + N)
    Npad = (N + 7) // 8 * 8
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9))
    wd = ho.pack_weight_dgrad(w.cuda(), Npad)
    segs, r = [], 0
    for h, w_ in sizes:
        segs.append(ho.Seg(B, h, w_, r))
        r += B * h * w_
    dzs = [bf(torch.randn(B, N, h, w_, generator=g)) for h, w_ in sizes]
    acts = [bf(torch.randn(B, C, h, w_, generator=g)) for h, w_ in sizes]         # the producer's ReLU output (mask = act > 0)
    dz_rows = torch.zeros(r, Npad)
    dz_rows[:, :N] = torch.cat([nhwc_rows(d) for d in dzs])
    dz_rows = dz_rows.cuda().bfloat16()
    mask = torch.cat([nhwc_rows(a) for a in acts]).cuda().bfloat16()
    outs = []
    keep_all = ho.HALO_ALL
    for halo in (True, False):
        ho.HALO_CONV, ho.HALO_ALL = halo, True
        try:
            cs = torch.zeros(C, device='cuda')
            dx = ho.conv2d_dgrad_rows(dz_rows, segs, segs, wd, C, 3, 3, 1, 1, 1, mask=mask, colsum=cs)
            outs.append((dx, cs))
        finally:
            ho.HALO_CONV, ho.HALO_ALL = True, keep_all
    torch.cuda.synchronize()
    ref_cs = torch.zeros(C)
    for dz, a, s_ in zip(dzs, acts, segs):
        x = torch.zeros(B, C, s_.H, s_.W, requires_grad=True)
        F.conv2d(x, w, None, 1, 1).backward(dz)
        ref = x.grad * (a > 0)
        ref_cs += ref.sum((0, 2, 3))
        sc = float(ref.abs().max())
        assert close(ho.rows_to_nchw(outs[0][0], s_), ref, 1e-2, 1e-2 * sc)
    assert close(outs[0][1], ref_cs, 1e-2, 1e-2 * float(ref_cs.abs().max()))
    assert close(outs[0][0], outs[1][0], 1e-2, 1e-2 * sc) and close(outs[0][1], outs[1][1], 1e-3, 1e-3 * float(ref_cs.abs().max()))


@pytest.mark.parametrize('N,out_f32', [(256, False), (180, True), (9, True), (72, False)])
def test_conv_split_k_matches_the_direct_kernel(ho, N, out_f32):
    """Small-output / deep-K convolutions run split-K (aod_conv2d_ws: fp32 partial sums in a workspace + a finalize pass).  Same
    epilogue semantics as the direct kernel on a two-segment input: bias, residual, mask, ReLU, column sums, fp32 / bf16 / N tails."""
    import ctypes as C
    from aod_meh_hua_amd._C import call, lib, ptr, stream
    B, Cin = 2, 1024
    sizes = [(6, 5), (3, 3)]
    g = synth.gen(300 + N)
    segs, r = [], 0
    for h, w_ in sizes:
        segs.append(ho.Seg(B, h, w_, r)); r += B * h * w_
    x = torch.randn(r, Cin, generator=g).cuda().bfloat16()
    w = (torch.randn(N, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)).cuda()
    wp = ho.pack_weight_fwd(w)
    bias = torch.randn(N, generator=g).cuda()
    res = None if out_f32 else torch.randn(r, N, generator=g).cuda().bfloat16()
    mask = None if out_f32 else torch.randn(r, N, generator=g).cuda().bfloat16()
    d = ho.make_desc(Cin, N, 3, 3, 1, 1, 1, segs, segs, False, True, out_f32)
    need = lib.aod_conv2d_ws_bytes(C.byref(d))
    assert need > 0 and need % (r * N * 4) == 0, 'the heuristic should pick split-K for tiny images x 144 K-steps'
    outs = []
    for use_ws in (False, True):
        y = torch.zeros(r, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device='cuda')
        cs = torch.zeros(N, device='cuda')
        ws = torch.full((need // 4,), 7.0, device='cuda') if use_ws else None          # (dirty on purpose: it need not be initialised)
        call('aod_conv2d_ws', C.byref(d), ptr(x), ptr(wp), ptr(y), None, ptr(bias), ptr(res), ptr(mask), None, None, ptr(cs),
             ptr(ws), need if use_ws else 0, stream())
        torch.cuda.synchronize()
        outs.append((y.float().cpu(), cs.cpu()))
    (y0, c0), (y1, c1) = outs
    assert float(y0.abs().max()) > 0.5
    assert torch.allclose(y1, y0, rtol=1e-2 if not out_f32 else 1e-5, atol=1e-2 if not out_f32 else 1e-5)
    assert torch.allclose(c1, c0, rtol=1e-3, atol=2e-2 * (1 if out_f32 else 4))
    # too small a workspace is an argument error, not a launch
    ws = torch.zeros(16, device='cuda')
    rc = lib.aod_conv2d_ws(C.byref(d), ptr(x), ptr(wp), ptr(y), None, None, None, None, None, None, None, ptr(ws), 64, stream())
    assert rc == -1 and b'workspace' in lib.aod_last_error()


@pytest.mark.parametrize('sizes', [[(17, 19)], [(16, 16), (9, 7)], [(5, 4), (3, 3), (2, 1)]])
def test_conv_dgrad_stride2_class_major_with_fused_activation_backward(ho, sizes):
    """dgrad of a 3x3 stride-2 conv runs class-major (rows grouped by the parity class of the destination pixel, dead taps skipped,
    destination rows scattered): odd sizes, several segments, and the fused epilogue (residual gradient + ReLU mask + column sums)
    against autograd on the same bf16 operands."""
    B, C, N = 2, 128, 64
    g = synth.gen(41 + len(sizes))
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9))
    xs_, zs_, xr, zr = [], [], 0, 0
    dz_rows, want, res_rows, mask_rows = [], [], [], []
    for (H, W) in sizes:
        x = bf(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
        y = F.conv2d(x, w, None, 2, 1, 1)
        dz = bf(torch.randn(y.shape, generator=g))
        y.backward(dz)
        res = bf(torch.randn(B, C, H, W, generator=g))
        mask = bf(torch.randn(B, C, H, W, generator=g))
        OH, OW = y.shape[-2:]
        xs_.append(ho.Seg(B, H, W, xr)); zs_.append(ho.Seg(B, OH, OW, zr))
        xr += B * H * W; zr += B * OH * OW
        dz_rows.append(nhwc_rows(dz)); res_rows.append(nhwc_rows(res)); mask_rows.append(nhwc_rows(mask))
        want.append(nhwc_rows((x.grad + res) * (mask > 0)))
    dz_rows = torch.cat(dz_rows).cuda().bfloat16()
    res_rows, mask_rows, want = torch.cat(res_rows).cuda().bfloat16(), torch.cat(mask_rows).cuda().bfloat16(), torch.cat(want)
    wd = ho.pack_weight_dgrad(w.cuda(), N)
    cs = torch.zeros(C, device='cuda')
    dx = ho.conv2d_dgrad_rows(dz_rows, zs_, xs_, wd, C, 3, 3, 2, 1, 1, res=res_rows, mask=mask_rows, colsum=cs)
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert close(dx, want, 1e-2, 1e-2 * scale)
    assert close(cs, dx.float().sum(0), 1e-3, 5e-2 * scale)           # column sums of the fp32 values behind what was stored (bf16)
    assert close(cs, want.sum(0), 2e-2, 0.3 * scale)


@pytest.mark.parametrize('prec', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('shape', [(2, 33, 20, 128, 256), (3, 16, 16, 256, 512), (1, 7, 130, 64, 128)])
def test_inplace_pointwise_stride2_dgrad_is_a_lattice_launch(ho, prec, shape):
    """dX += conv_T(dZ, W) of a 1x1 / stride-2 conv with res == dst (the running sum of a gradient junction, functional.GradAcc): the library
    runs a GEMM over the dZ pixels and stores its rows at the (even, even) pixels of dX (conv.hip, lattice launch) instead of a transposed
    launch over all of dX.  Against the general launch into a separate destination: bf16 -- identical bits everywhere; reference-precision
    mode -- identical on the lattice, and off it the untouched sum against its re-rounded copy (head + tail re-split)."""
    from aod_meh_hua_amd import functional as AF
    AF.set_precision(prec)
    try:
        B, H, W, I, O = shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        g = torch.Generator(device='cuda').manual_seed(7)
        rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
        w = rnd(O, I, 1, 1) / np.sqrt(O)
        dz_f, part_f = rnd(B * OH * OW, O), rnd(B * H * W, I)
        if prec == 'bf16x3':
            dz, part = ho.x3_split(dz_f), ho.x3_split(part_f)
            wd = ho.x3_split(w.reshape(O, I).t().contiguous()).view(I, 1, 1, -1)
        else:
            dz, part = dz_f.bfloat16(), part_f.bfloat16()
            wd = ho.pack_weight_dgrad(w, O)
        zs, xs = [ho.Seg(B, OH, OW, 0)], [ho.Seg(B, H, W, 0)]
        want = ho.conv2d_dgrad_rows(dz, zs, xs, wd, I, 1, 1, 2, 0, 1, res=part)
        assert want.data_ptr() != part.data_ptr()
        acc = part.clone()
        got = ho.conv2d_dgrad_rows(dz, zs, xs, wd, I, 1, 1, 2, 0, 1, res=acc, out=acc)
        torch.cuda.synchronize()
        assert got.data_ptr() == acc.data_ptr()
        lat = torch.zeros(B, H, W, dtype=torch.bool, device='cuda')
        lat[:, ::2, ::2] = True
        lat = lat.reshape(-1)
        assert torch.equal(got[lat], want[lat])
        assert torch.equal(got[~lat], part[~lat])
        if prec == 'bf16':
            assert torch.equal(got, want)
        else:
            assert float((ho.x3_merge(got) - ho.x3_merge(want)).abs().max()) <= 2e-5 * float(part_f.abs().max())
        # and the values: fp32 reference on the same operands
        ref = part_f.view(B, H, W, I).clone()
        ref[:, ::2, ::2] += (dz_f @ w.reshape(O, I)).view(B, OH, OW, I)
        out = ho.x3_merge(got) if prec == 'bf16x3' else got.float()
        assert float((out - ref.reshape(-1, I)).abs().max()) < (1e-4 if prec == 'bf16x3' else 6e-2) * float(ref.abs().max())
    finally:
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_dgrad_and_wgrad(ho, case):
    B, C, H, W, N, R, stride, pad, dil = case
    if C == 8:
        Creal = 3
    else:
        Creal = C
    g = synth.gen(200 + N + C)
    x = bf(torch.randn(B, Creal, H, W, generator=g)).requires_grad_(True)
    w = bf(torch.randn(N, Creal, R, R, generator=g) / np.sqrt(Creal * R * R)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad, dil)
    dz = bf(torch.randn(y.shape, generator=g))
    y.backward(dz)
    OH, OW = y.shape[-2:]
    Npad = (N + 7) // 8 * 8
    dz_rows = torch.zeros(B * OH * OW, Npad)
    dz_rows[:, :N] = nhwc_rows(dz)
    dz_rows = dz_rows.cuda().bfloat16()
    xpad = torch.zeros(B, C, H, W)
    xpad[:, :Creal] = x.detach()
    x_rows = nhwc_rows(xpad).cuda().bfloat16()
    xs, zs = [ho.Seg(B, H, W)], [ho.Seg(B, OH, OW)]
    # wgrad: slab form (default for <= 9 taps: every pixel split stores its partial, the unpack adds the slabs in order) ...
    dw = ho.conv2d_wgrad_rows(x_rows, xs, dz_rows, zs, R, R, stride, pad, dil)
    gw = ho.unpack_wgrad(dw, N, Creal)
    torch.cuda.synchronize()
    scale = float(w.grad.abs().max())
    assert close(gw, w.grad, 5e-3, 5e-3 * scale)
    if R * R <= 9:
        assert dw.dim() == 5
        gw2 = ho.unpack_wgrad(ho.conv2d_wgrad_rows(x_rows, xs, dz_rows, zs, R, R, stride, pad, dil), N, Creal)
        assert torch.equal(gw, gw2)                         # ... is bit-reproducible run to run
        # ... and agrees with the fp32-atomic accumulation form of the same kernel
        acc = torch.zeros(Npad, R, R, C, device='cuda')
        ho.conv2d_wgrad_rows(x_rows, xs, dz_rows, zs, R, R, stride, pad, dil, dw=acc)
        gw3 = ho.unpack_wgrad(acc, N, Creal, clear=False)
        assert close(gw3, gw, 1e-4, 1e-5 * scale)
    # dgrad
    wd = ho.pack_weight_dgrad(w.detach().cuda(), Npad)
    wd_full = wd
    if Creal != C:   # pad the "input channel" axis of the dgrad packing so that dX has C columns
        wd_full = torch.zeros(C, R, R, Npad, dtype=torch.bfloat16, device='cuda')
        wd_full[:Creal] = wd
    dx = ho.conv2d_dgrad_rows(dz_rows, zs, xs, wd_full, C, R, R, stride, pad, dil)
    torch.cuda.synchronize()
    got = rows_nchw(dx, B, H, W)[:, :Creal]
    assert close(got, x.grad, 1e-2, 1e-2 * float(x.grad.abs().max()))


@pytest.mark.parametrize('N', [9, 36, 180, 720])
def test_pad_cast_colsum(ho, N):
    """Prediction-conv gradient: fp32 [M, N] -> bf16 [M, Npad] (zero pad columns, optional fused ReLU mask) + fp32 column sums."""
    M = 1237
    g = synth.gen(60 + N)
    x = torch.randn(M, N, generator=g)
    a = torch.randn(M, N, generator=g)
    npad = (N + 7) // 8 * 8
    for relu_out in (None, a):
        dz, cs = ho.pad_cast_colsum(x.cuda(), npad, relu_out.cuda() if relu_out is not None else None)
        torch.cuda.synchronize()
        want = x if relu_out is None else x * (a > 0)
        assert dz.shape == (M, npad) and dz.dtype == torch.bfloat16
        assert torch.equal(dz[:, :N].float().cpu(), want.to(torch.bfloat16).float())
        assert float(dz[:, N:].float().abs().max()) == 0.0 if npad > N else True
        assert close(cs[:N], want.sum(0), 1e-4, 1e-3)


def test_elementwise_ops(ho):
    g = synth.gen(5)
    B, C, H, W = 2, 64, 13, 11
    x = bf(torch.randn(B, C, H, W, generator=g))
    xr = nhwc_rows(x).cuda().bfloat16()
    y, s = ho.maxpool3x3s2(xr, ho.Seg(B, H, W))
    assert torch.equal(rows_nchw(y, B, s.H, s.W).float().cpu(), F.max_pool2d(x, 3, 2, 1))
    img = torch.randn(2, 3, 10, 12, generator=g)
    r, sg = ho.nchw_to_rows(img.cuda(), 8)
    back = rows_nchw(r, 2, 10, 12).float().cpu()
    assert torch.equal(back[:, :3], bf(img)) and (back[:, 3:] == 0).all()
    # upsample-add and adjoint (exact 2x and the ragged 7 -> 13 case of F.interpolate(size=...))
    for (h, w, Hh, Ww) in ((4, 5, 8, 10), (7, 6, 13, 11)):
        src = bf(torch.randn(B, C, h, w, generator=g))
        dst = bf(torch.randn(B, C, Hh, Ww, generator=g))
        ref = bf(dst + F.interpolate(src, size=(Hh, Ww), mode='nearest'))
        d = nhwc_rows(dst).cuda().bfloat16()
        ho.upsample_add_(d, ho.Seg(B, Hh, Ww), nhwc_rows(src).cuda().bfloat16(), ho.Seg(B, h, w))
        assert torch.equal(rows_nchw(d, B, Hh, Ww).float().cpu(), ref)
        lat = nhwc_rows(dst).cuda().bfloat16()
        o2 = ho.upsample_add(lat, ho.Seg(B, Hh, Ww), nhwc_rows(src).cuda().bfloat16(), ho.Seg(B, h, w))       # out of place: the lateral is untouched
        assert torch.equal(o2, d) and torch.equal(lat.float().cpu(), nhwc_rows(dst))
        gd = bf(torch.randn(B, C, Hh, Ww, generator=g))
        s_ = src.clone().requires_grad_(True)
        F.interpolate(s_, size=(Hh, Ww), mode='nearest').backward(gd)
        gs = torch.zeros(B * h * w, C, device='cuda', dtype=torch.bfloat16)
        ho.upsample_add_bwd_(gs, ho.Seg(B, h, w), nhwc_rows(gd).cuda().bfloat16(), ho.Seg(B, Hh, Ww))
        assert close(rows_nchw(gs, B, h, w), s_.grad, 1e-2, 2e-2)
        assert torch.equal(ho.upsample_add_bwd(nhwc_rows(gd).cuda().bfloat16(), ho.Seg(B, Hh, Ww), ho.Seg(B, h, w)), gs)      # written, not accumulated
    a, b = bf(torch.randn(1000, 64, generator=g)), bf(torch.randn(1000, 64, generator=g))
    assert torch.equal(ho.add_relu(a.cuda().bfloat16(), b.cuda().bfloat16()).float().cpu(), bf(F.relu(a + b)))


def test_act_bwd_bn_and_bias(ho):
    g = synth.gen(6)
    M, N = 1500, 320
    z = bf(torch.randn(M, N, generator=g))
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    mean, var = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    zz = z.clone().requires_grad_(True)
    gm_, bt_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.relu((zz - mean) * invstd * gm_ + bt_)
    gout = bf(torch.randn(M, N, generator=g))
    a.backward(gout)
    dz, gmk, dbeta, dgamma = ho.act_bwd(gout.cuda().bfloat16(), bf(a.detach()).cuda().bfloat16(), z.cuda().bfloat16(),
                                        (gamma * invstd).cuda(), mean.cuda(), invstd.cuda(), relu=True, want_gm=True)
    torch.cuda.synchronize()
    assert close(dz, zz.grad, 1e-2, 1e-2)
    assert close(dbeta, bt_.grad, 1e-3, 1e-2) and close(dgamma, gm_.grad, 1e-3, 2e-2)
    assert close(gmk, gout * (a.detach() > 0), 1e-2, 1e-3)
    # bias-only mode with fp32 upstream gradient, no relu
    g32 = torch.randn(M, N, generator=g)
    dz2, _, db2, dg2 = ho.act_bwd(g32.cuda(), relu=False)
    assert dg2 is None and close(db2, g32.sum(0), 1e-4, 1e-3) and close(dz2, g32, 1e-2, 1e-3)


def test_losses_match_reference_golden(ho):
    gold = np.load(os.path.join(G, 'losses.npz'))
    li = synth.loss_inputs()
    n = li['num_total_samples']
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in li.items()}
    noR, sums = ho.edl_focal_l1_fwd(dev['logits'], dev['labels'], dev['label_weights'], dev['bbox_pred'], dev['bbox_targets'], dev['bbox_weights'])
    torch.cuda.synchronize()
    assert np.allclose(noR.cpu().numpy(), gold['loss_noR'], rtol=2e-5, atol=1e-7)
    s = sums.cpu().numpy()
    assert np.allclose(s[0] / n, gold['loss_cls'], rtol=1e-5) and np.allclose(s[1] / n, gold['loss_bbox'], rtol=1e-5)
    assert np.allclose(s[2] / 1024, gold['loss_noR'].mean(), rtol=1e-5)
    one = torch.full((1,), 1.0 / n, device='cuda')
    gc, gb = ho.edl_focal_l1_bwd(dev['logits'], dev['labels'], dev['label_weights'], dev['bbox_pred'], dev['bbox_targets'], dev['bbox_weights'],
                                 one, one, None, 1.0 / 1024)
    assert np.allclose(gc.cpu().numpy(), gold['grad_logits'], rtol=5e-4, atol=2e-7)
    assert np.array_equal(gb.cpu().numpy(), gold['grad_bbox'])
    # padded bf16 dZ layout: A = 4 anchors per pixel, pitch 88 (>= 4*20), bbox pitch 16
    gc2, gb2 = ho.edl_focal_l1_bwd(dev['logits'], dev['labels'], dev['label_weights'], dev['bbox_pred'], dev['bbox_targets'], dev['bbox_weights'],
                                   one, one, None, 1.0 / 1024, out_bf16=True, A=4, pitch_cls=88, pitch_box=16)
    ref = torch.from_numpy(gold['grad_logits']).view(256, 80)
    assert close(gc2[:, :80], ref, 1e-2, 1e-6) and (gc2[:, 80:] == 0).all()
    # MEH
    lam = dev['lam']
    out = ho.meh_loss_fwd(lam, noR, dev['bbox_weights'])
    assert np.allclose(out.item() / 1024 * 5, gold['loss_L'], rtol=1e-5)
    gl = ho.meh_loss_bwd(lam, noR, dev['bbox_weights'], torch.full((1,), 5.0 / 1024, device='cuda'))
    assert np.allclose(gl.cpu().numpy().reshape(-1), gold['grad_lam'], rtol=1e-4, atol=1e-9)


def test_edl_module_reduction_none_is_elementwise_like_the_reference(ho):
    """EDL_Softmax_FocalLoss.forward(reduction='none') returns the [N, C] elementwise loss (EDL_Softmax_FocalLoss.py:51-69): its class sum is
    the reference's golden loss_noR (tools/golden/make_golden.py: `head.loss_cls(x, labels, reduction_override='none').sum(-1)`), values and
    gradients match the oracle's autograd, and the weighted / averaged reductions agree with the fused row kernel."""
    from aod_meh_hua_amd.models.losses.edl_softmax_focal_loss import EDL_Softmax_FocalLoss
    gold = np.load(os.path.join(G, 'losses.npz'))
    li = synth.loss_inputs()
    mod = EDL_Softmax_FocalLoss(num_classes=20, annealing_step=1, last_activation='relu')
    x = li['logits'].cuda().requires_grad_(True)
    lab = li['labels'].cuda()
    el = mod(x, lab, reduction_override='none')
    assert el.shape == (1024, 20)
    assert np.allclose(el.sum(-1).detach().cpu().numpy(), gold['loss_noR'], rtol=2e-5, atol=1e-7)
    xr = li['logits'].clone().requires_grad_(True)
    ref = olosses.edl_softmax_focal_none(xr, li['labels'])
    assert close(el.detach(), ref.detach(), 2e-5, 1e-7)
    g = torch.rand(1024, 20, generator=synth.gen(3))
    (el * g.cuda()).sum().backward()
    (ref * g).sum().backward()
    assert close(x.grad, xr.grad, 1e-3, 1e-7)
    # per-row weights + avg_factor (the reference's loss_cls call, Lambda_L2.py:118): fused row kernel == elementwise kernel
    w = li['label_weights'].cuda()
    a = mod(x, lab, w, avg_factor=li['num_total_samples'])
    b = (mod(x, lab, reduction_override='none') * w[:, None]).sum() / li['num_total_samples']
    assert np.allclose(float(a), float(b), rtol=1e-5) and np.allclose(float(a), gold['loss_cls'], rtol=1e-5)
    we = torch.rand(1024, 20, generator=synth.gen(4)).cuda()          # per-element weights (weight_reduce_loss, losses/utils.py:28-54)
    c = mod(x, lab, we, reduction_override='sum')
    assert np.allclose(float(c), float((ref.detach() * we.cpu()).sum()), rtol=1e-4)


def test_losses_edge_cases(ho):
    """All-background rows, extreme logits (softmax saturation -> clamps), zero rows."""
    x = torch.tensor([[50., -50.] + [0.] * 18, [-80.] * 19 + [80.], [0.] * 20], device='cuda')
    lab = torch.tensor([0, 20, 19], device='cuda')
    lw = torch.ones(3, device='cuda')
    noR, _ = ho.edl_focal_l1_fwd(x, lab, lw)
    ref = olosses.edl_softmax_focal_none(x.cpu(), lab.cpu()).sum(-1)
    assert torch.isfinite(noR).all() and close(noR, ref, 1e-4, 1e-6)
    e, s = ho.edl_focal_l1_fwd(torch.zeros(0, 20, device='cuda'), torch.zeros(0, dtype=torch.long, device='cuda'), torch.zeros(0, device='cuda'))
    assert e.numel() == 0 and float(s.sum()) == 0


def test_cpu_tensor_is_refused(ho):
    from aod_meh_hua_amd._C import AodHipError
    with pytest.raises(AodHipError):
        ho.add_relu(torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16))


def test_conv_256x256_tile_equals_the_default_tile(tmp_path):
    """The 256 x 256 tile (two epilogue passes, no operand prefetch; chosen automatically when its tiles fill whole rounds of the CUs)
    against the 128 x 128 tile on a ragged, five-segment tower shape, forward and dgrad with mask + column sums: same K order per
    output element, so the bf16 results are identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ('0', '1'):
        f = str(tmp_path / f'tile{flag}.pt')
        p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dbg', 'tile256_check.py'), f], env=dict(os.environ, AOD_TILE_256=flag),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[flag] = torch.load(f)
    assert torch.equal(outs['0']['y'], outs['1']['y'])
    assert torch.equal(outs['0']['dx'], outs['1']['dx'])
    assert torch.allclose(outs['0']['cs'], outs['1']['cs'], rtol=1e-4, atol=1e-2)          # (atomic column sums: order differs)
    assert float(outs['1']['y'].float().abs().mean()) > 0.1 and float(outs['1']['dx'].float().abs().mean()) > 0.01


def test_wgrad_256x256_tile_against_fp32_reference(ho):
    """The 256 x 256 wgrad tile (chosen for N, K multiples of 256 and >= 49152 pixels: the head towers at the bench size) on a ragged
    two-segment shape against torch's fp32 weight gradient of the same bf16-rounded operands, and bit-reproducible run to run."""
    g = synth.gen(77)
    levels = [(12, 64, 64), (3, 23, 19)]
    C = N = 256
    w = bf(torch.randn(N, C, 3, 3, generator=g) / np.sqrt(C * 9.0))
    x_rows, dz_rows, xs, zs, want, r0 = [], [], [], [], torch.zeros(N, C, 3, 3), 0
    for B, H, W in levels:
        x = bf(torch.randn(B, C, H, W, generator=g))
        dz = bf(torch.randn(B, N, H, W, generator=g))
        want += torch.nn.grad.conv2d_weight(x, w.shape, dz, stride=1, padding=1)
        xs.append(ho.Seg(B, H, W, r0)); zs.append(ho.Seg(B, H, W, r0)); r0 += B * H * W
        x_rows.append(nhwc_rows(x)); dz_rows.append(nhwc_rows(dz))
    assert r0 >= 49152
    x_rows, dz_rows = torch.cat(x_rows).cuda().bfloat16(), torch.cat(dz_rows).cuda().bfloat16()
    dw = ho.conv2d_wgrad_rows(x_rows, xs, dz_rows, zs, 3, 3, 1, 1, 1)
    assert dw.dim() == 5 and dw.shape[0] > 14               # slabs of the big tile (256 / 9 tiles = 28 pixel splits)
    gw = ho.unpack_wgrad(dw, N, C)
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert close(gw, want, 5e-3, 5e-3 * scale)
    gw2 = ho.unpack_wgrad(ho.conv2d_wgrad_rows(x_rows, xs, dz_rows, zs, 3, 3, 1, 1, 1), N, C)
    assert torch.equal(gw, gw2)


@pytest.mark.parametrize('shape', [(3, 37, 29, 64, 192, True, True), (2, 64, 64, 256, 64, False, True), (1, 50, 41, 128, 256, True, False),
                                   (2, 16, 16, 1024, 128, True, True)])
def test_pointwise_streaming_kernel_equals_the_general_kernel(ho, shape):
    """csrc/pointwise.hip (persistent streaming GEMM for 1x1 / stride-1 convs) against the implicit-GEMM kernel on the same launch:
    scale + shift + residual + mask + ReLU epilogue and column sums; same K order per output element -> identical bf16 results."""
    from aod_meh_hua_amd._C import lib
    B, H, W, Ci, Co, with_res, with_mask = shape
    g = synth.gen(300 + Ci + Co)
    M = B * H * W
    segs = [ho.Seg(B, H, W, 0)]
    x = torch.randn(M, Ci, generator=g).cuda().bfloat16()
    wp = ho.pack_weight_fwd((torch.randn(Co, Ci, 1, 1, generator=g) / np.sqrt(Ci)).cuda())
    scale, shift = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    res = torch.randn(M, Co, generator=g).cuda().bfloat16() if with_res else None
    mask = torch.randn(M, Co, generator=g).cuda().bfloat16() if with_mask else None
    outs = []
    prev = lib.aod_set_pointwise_mode(0)
    try:
        for mode in (0, 1):
            lib.aod_set_pointwise_mode(mode)
            out = torch.full((M, Co), 7.0, device='cuda', dtype=torch.bfloat16)
            cs = torch.zeros(Co, device='cuda')
            d = ho.make_desc(Ci, Co, 1, 1, 1, 0, 1, segs, segs, False, True, False)
            ho.call('aod_conv2d', ho.C.byref(d), ho.ptr(x), ho.ptr(wp), ho.ptr(out), ho.ptr(scale), ho.ptr(shift), ho.ptr(res), ho.ptr(mask), None, None,
                    ho.ptr(cs), ho.stream())
            torch.cuda.synchronize()
            outs.append((out, cs))
    finally:
        lib.aod_set_pointwise_mode(prev)
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert float(outs[1][0].float().abs().mean()) > 0.05
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-2)


def test_three_stage_ring_equals_the_two_stage_loop(tmp_path):
    """The small 4-wave conv tiles keep two LDS stages in flight (counted s_waitcnt vmcnt + raw s_barrier instead of __syncthreads();
    AOD_RING3=0 restores the two-stage loop): same K order per output element, so every result is identical -- forward with BN, residual and
    ReLU, stride 2, ragged N, dgrad with mask; K from 1 to 72 K-steps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ('0', '1'):
        f = str(tmp_path / f'ring{flag}.pt')
        p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dbg', 'ring3_check.py'), f], env=dict(os.environ, AOD_RING3=flag),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[flag] = torch.load(f)
    assert set(outs['0']) == set(outs['1']) and len(outs['0']) >= 12
    for k in outs['0']:
        assert torch.equal(outs['0'][k], outs['1'][k]), k
        assert float(outs['1'][k].float().abs().mean()) > 1e-3, k
