"""Thin functional layer over the C ABI (no autograd here): shape bookkeeping + pointer passing.

Activations are "row tensors": a flat [rows, C] bf16 buffer holding one or more NHWC blocks
(`Seg` records).  A single-image-batch tensor [B, C, H, W] in torch.channels_last memory format
is the same bytes as rows [B*H*W, C]."""
import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _C
from ._C import ConvDesc, ConvSeg, call, lib, ptr, stream


# Reference-precision mode (functional.set_precision('bf16x3'), csrc/conv.hip "X3"): bf16 activations / gradients are X-LAYOUT rows -- a
# tensor of c logical channels has xw(c) = 2 * ceil32(c) columns [h(0..31) | l(0..31) | h(32..63) | ...], value = head + tail -- and the
# descriptors carry x3 = 1.  The wrappers below take and return such rows when X3 is set; channel COUNTS passed to them (N, Cin) stay logical.
X3 = False
DETERMINISTIC = False       # aod_set_deterministic (functional.set_deterministic): ordered column sums


def set_deterministic(on=True):
    global DETERMINISTIC
    rc = lib.aod_set_deterministic(1 if on else 0)
    if rc < 0:
        raise RuntimeError(lib.aod_last_error().decode())
    DETERMINISTIC = bool(on)


def xw(c):
    """physical width (bf16 columns) of an X-layout row with c logical channels"""
    return 2 * ((c + 31) // 32 * 32)


def width(c):
    """row width of a bf16 activation tensor with c channels in the current precision mode"""
    return xw(c) if X3 else c


_ROW_TABLES = {}     # wgrad row tables, one per conv geometry (shared by every layer / iteration with that geometry)


# ---- optional per-launch profiling (bench.py): (kind, tile, flops, start_event, end_event, scope) on the CURRENT stream
PROFILE = None
SCOPE = 'head'          # which part of the network the launches belong to ('backbone' | 'neck' | 'head'): the detector sets it around its
                        # sub-modules and a conv's backward restores the scope of its forward -- only read while PROFILE is on


class scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global SCOPE
        self.prev, SCOPE = SCOPE, self.name

    def __exit__(self, *exc):
        global SCOPE
        SCOPE = self.prev
        return False


def _prof(kind, desc, fn, alg=None):
    """alg = (Cin, Cout) of the layer as the reference defines it: channel pads (stem 3 -> 8, prediction convs 180 -> 184 ...) are
    excluded from the algorithmic FLOPs."""
    if PROFILE is None:
        return fn()
    # algorithmic FLOPs = 2 * (forward output pixels) * Cout * R*S*Cin, for dgrad too (dZ pixels = seg.H*seg.W)
    if desc.transposed:
        m = sum(desc.seg[i].B * desc.seg[i].H * desc.seg[i].W for i in range(desc.nseg))
    else:
        m = sum(desc.seg[i].B * desc.seg[i].OH * desc.seg[i].OW for i in range(desc.nseg))
    cn = alg[0] * alg[1] if alg is not None else desc.N * desc.C
    flops = 2.0 * m * desc.R * desc.S * cn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((kind, (m, desc.N, desc.R * desc.S * desc.C, desc.R * desc.S, desc.stride), flops, e0, e1, SCOPE))
    return r


def prof_flops(kind, shape, flops, fn):
    """like _prof for launches that are not one convolution (fused bottleneck): `shape` = (M, N, K, taps, stride) of the listing"""
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((kind, shape, float(flops), e0, e1, SCOPE))
    return r


BYTES_PROFILE = None     # bench.py: (kernel name, algorithmic bytes, start_event, end_event) for the HBM-bound row kernels


def prof_bytes(name, nbytes, fn):
    if BYTES_PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    BYTES_PROFILE.append((name, float(nbytes), e0, e1))
    return r


@dataclass(frozen=True)
class Seg:
    B: int
    H: int
    W: int
    row0: int = 0

    @property
    def rows(self):
        return self.B * self.H * self.W


def out_hw(h, w, R, S, stride, pad, dil):
    return ((h + 2 * pad - dil * (R - 1) - 1) // stride + 1, (w + 2 * pad - dil * (S - 1) - 1) // stride + 1)


def out_segs(segs: Sequence[Seg], R, S, stride, pad, dil) -> List[Seg]:
    out, r = [], 0
    for s in segs:
        oh, ow = out_hw(s.H, s.W, R, S, stride, pad, dil)
        out.append(Seg(s.B, oh, ow, r))
        r += s.B * oh * ow
    return out


def make_desc(C_, N, R, S, stride, pad, dil, src_segs: Sequence[Seg], dst_segs: Sequence[Seg], transposed=False,
              relu=False, out_f32=False) -> ConvDesc:
    d = ConvDesc()
    d.C, d.N, d.R, d.S, d.stride, d.pad, d.dil = C_, N, R, S, stride, pad, dil
    d.transposed, d.relu, d.out_f32, d.nseg, d.x3 = int(transposed), int(relu), int(out_f32), len(src_segs), int(X3)
    assert 1 <= len(src_segs) <= 8 and len(src_segs) == len(dst_segs)
    for i, (s, o) in enumerate(zip(src_segs, dst_segs)):
        d.seg[i] = ConvSeg(s.B, s.H, s.W, o.H, o.W, s.row0, o.row0)
    return d


def pack_weight_fwd(w_oihw: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    O, I, R, S = w_oihw.shape
    cpad = cpad or (I + 7) // 8 * 8
    out = torch.empty(O, R, S, cpad, dtype=torch.bfloat16, device=w_oihw.device)
    call('aod_pack_weight_fwd', ptr(w_oihw.contiguous()), ptr(out), O, I, R, S, cpad, stream())
    return out


def pack_weight_dgrad(w_oihw: torch.Tensor, opad: Optional[int] = None, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`scale` fp32 [O]: folded eval-BN scale of the conv's own BatchNorm (dX = conv_T(gm, scale*W))."""
    O, I, R, S = w_oihw.shape
    opad = opad or (O + 7) // 8 * 8
    out = torch.empty(I, R, S, opad, dtype=torch.bfloat16, device=w_oihw.device)
    call('aod_pack_weight_dgrad', ptr(w_oihw.contiguous()), ptr(out), O, I, R, S, opad, ptr(scale), stream())
    return out


SPLITK = os.environ.get('AOD_SPLITK', '1') != '0'          # debug switch: 0 = never hand a split-K workspace to the library


def _splitk_ws(d, device):
    """(workspace, bytes) for aod_conv2d_ws: a scratch tensor from torch's allocator when the library's heuristic wants split-K for this
    descriptor (graph-safe: under capture it comes from the graph's pool), else (None, 0)."""
    if not SPLITK:
        return None, 0
    n = lib.aod_conv2d_ws_bytes(C.byref(d))
    if n == 0:
        return None, 0
    return torch.empty(n // 4, dtype=torch.float32, device=device), n


HALO_CONV = os.environ.get('AOD_HALO_CONV', '1') != '0'   # debug switch: 0 = the narrow 3x3 convs (prediction heads) go through the general kernel
# Which launches take the halo-tile kernel (csrc/halo_conv.hip).  Measured on MI355X at 16 x 512^2 (profiles/r03_halo_conv_ab.txt): it wins on
# the INPUT-bound forward convs (retina_reg 67.6 -> 60 us, retina_L 62.9 -> 45 us: the 10 x 18 halo crosses HBM once instead of 7.8 times) and
# only draws level on retina_cls (N = 180: 128-133 vs 128-143 us) and loses on the dgrads (N = 256 outputs: 145-187 vs 139 us, 79-94 vs
# 60 us): at one workgroup per CU its 24-MFMA K-steps are too short for two waves per SIMD.  AOD_HALO_CONV=all routes every qualifying
# launch (the kernel tests do), the default takes the forward launches with at most 64 output channels.
HALO_ALL = os.environ.get('AOD_HALO_CONV', '1') == 'all'


def _halo_applies(d, narrow, *unsupported, dgrad=False):
    """the halo-tile kernel (aod_halo_conv3x3) takes 3x3 / stride-1 / pad-1 convs whose narrow side has < 256 channels and that need none
    of the epilogue operands it does not implement"""
    if not HALO_CONV or X3 or narrow >= 256 or any(u is not None for u in unsupported):
        return False
    if not HALO_ALL and (dgrad or narrow > 64):
        return False
    return bool(lib.aod_halo_conv3x3_applies(C.byref(d)))


def conv2d_rows(x_rows, src_segs, w_packed, N, R, S, stride=1, pad=0, dil=1, *, pre_scale=None, pre_shift=None,
                res=None, mask=None, post_scale=None, relu=False, out_f32=False, save_z=False, out=None,
                dst_segs=None, out_rows=None, alg=None):
    """Forward conv on row tensors.  Returns (y_rows, dst_segs[, z_rows])."""
    Cin = x_rows.shape[1]
    dst_segs = dst_segs or out_segs(src_segs, R, S, stride, pad, dil)
    rows = out_rows if out_rows is not None else sum(s.rows for s in dst_segs)
    if out is None:
        out = torch.empty(rows, N if out_f32 else width(N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x_rows.device)
    assert out.shape[1] == (N if out_f32 else width(N)), 'destination width does not match the precision mode'
    z = torch.empty(rows, N, dtype=torch.bfloat16, device=x_rows.device) if save_z else None
    d = make_desc(Cin, N, R, S, stride, pad, dil, src_segs, dst_segs, False, relu, out_f32)
    if R == 3 and not save_z and _halo_applies(d, N, pre_scale, res, mask, post_scale):
        _prof('fwd', d, lambda: call('aod_halo_conv3x3', C.byref(d), ptr(x_rows), ptr(w_packed), ptr(out), ptr(pre_shift), None, None, stream()), alg)
        return out, dst_segs
    # reference-precision mode: the narrow prediction convs (retina_reg 36, retina_L 9 fp32 output channels) on the chunk-resident halo kernel
    # (csrc/halo_x3.hip; AOD_HALO_X3=0: the general kernel, which computes 64 columns for them) -- identical bits
    if (X3 and R == 3 and out_f32 and N <= 48 and not save_z and pre_scale is None and res is None and mask is None and post_scale is None
            and os.environ.get('AOD_HALO_X3', '1') != '0' and lib.aod_halo_conv3x3_x3_applies(C.byref(d))):
        _prof('fwd', d, lambda: call('aod_halo_conv3x3_x3', C.byref(d), ptr(x_rows), ptr(w_packed), ptr(out), ptr(pre_shift), stream()), alg)
        return out, dst_segs
    ws, wsb = _splitk_ws(d, x_rows.device)
    _prof('fwd', d, lambda: call('aod_conv2d_ws', C.byref(d), ptr(x_rows), ptr(w_packed), ptr(out), ptr(pre_scale), ptr(pre_shift),
                                 ptr(res), ptr(mask), ptr(post_scale), ptr(z), None, ptr(ws), wsb, stream()), alg)
    return (out, dst_segs, z) if save_z else (out, dst_segs)


def conv2d_rows_grouped(xs, src_segs, ws, N, R, S, stride=1, pad=0, dil=1, *, pre_shifts=None, relu=False, outs=None, alg=None):
    """G forward convs of identical geometry in ONE launch (aod_conv2d_grouped): xs / ws / pre_shifts / outs are lists of G row tensors
    (inputs may be the same tensor), `src_segs` the shared segment list.  Returns (list of outputs, dst_segs)."""
    G = len(xs)
    Cin = xs[0].shape[1]
    dst_segs = out_segs(src_segs, R, S, stride, pad, dil)
    rows = sum(s.rows for s in dst_segs)
    if outs is None:
        outs = [torch.empty(rows, width(N), dtype=torch.bfloat16, device=xs[0].device) for _ in range(G)]
    d = make_desc(Cin, N, R, S, stride, pad, dil, src_segs, dst_segs, False, relu, False)
    arr = lambda ts: (C.c_void_p * G)(*[(t.data_ptr() if t is not None else None) for t in ts])
    for t in list(xs) + list(ws) + list(outs):
        ptr(t)                                   # (refuses CPU tensors)
    keep = [arr(xs), arr(ws), arr(outs), arr(pre_shifts) if pre_shifts is not None else None]
    if PROFILE is None:
        call('aod_conv2d_grouped', C.byref(d), G, keep[0], keep[1], keep[2], keep[3], None, None, stream())
    else:                                       # (profiling: the G groups count as G launches' worth of FLOPs in one record)
        for _ in range(1):
            _prof('fwd', d, lambda: call('aod_conv2d_grouped', C.byref(d), G, keep[0], keep[1], keep[2], keep[3], None, None, stream()),
                  (alg[0] * G, alg[1]) if alg is not None else (Cin * G, N))
    return outs, dst_segs


def conv2d_dgrad_rows_grouped(dzs, dz_segs, x_segs, wds, Cin, R, S, stride=1, pad=0, dil=1, *, masks=None, colsums=None, alg=None):
    """G dgrads of identical geometry in one launch: dX_g = conv_T(dZ_g, W_g) (* [mask_g > 0], column sums into colsums[g])."""
    G = len(dzs)
    Npad = dzs[0].shape[1]
    rows = sum(s.rows for s in x_segs)
    outs = [torch.empty(rows, width(Cin), dtype=torch.bfloat16, device=dzs[0].device) for _ in range(G)]
    d = make_desc(Npad, Cin, R, S, stride, pad, dil, dz_segs, x_segs, True, False, False)
    arr = lambda ts: (C.c_void_p * G)(*[(t.data_ptr() if t is not None else None) for t in ts])
    for t in list(dzs) + list(wds) + outs:
        ptr(t)
    keep = [arr(dzs), arr(wds), arr(outs), arr(masks) if masks is not None else None, arr(colsums) if colsums is not None else None]
    fn = lambda: call('aod_conv2d_grouped', C.byref(d), G, keep[0], keep[1], keep[2], None, keep[3], keep[4], stream())
    if PROFILE is None:
        fn()
    else:
        _prof('dgrad', d, fn, (alg[0] * G, alg[1]) if alg is not None else (Npad * G, Cin))
    return outs




def conv2d_dgrad_rows(dz_rows, dz_segs, x_segs, w_dgrad, Cin, R, S, stride=1, pad=0, dil=1, *, res=None, mask=None,
                      post_scale=None, out=None, x_rows_total=None, colsum=None, alg=None, out_f32=False):
    """dX = conv_transpose(dZ, W).  dz_rows [rows_out, Npad]; w_dgrad [Cin][R][S][Npad].
    res / mask / colsum: fused activation backward of the producer of x (see aod_conv2d)."""
    Npad = dz_rows.shape[1]
    rows = x_rows_total if x_rows_total is not None else sum(s.rows for s in x_segs)
    if out is None:
        out = torch.empty(rows, Cin if out_f32 else width(Cin), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dz_rows.device)
    d = make_desc(Npad, Cin, R, S, stride, pad, dil, dz_segs, x_segs, True, False, out_f32)
    if R == 3 and _halo_applies(d, Npad, res, post_scale, dgrad=True):
        _prof('dgrad', d, lambda: call('aod_halo_conv3x3', C.byref(d), ptr(dz_rows), ptr(w_dgrad), ptr(out), None, ptr(mask), ptr(colsum), stream()), alg)
        return out
    ws, wsb = _splitk_ws(d, dz_rows.device)
    _prof('dgrad', d, lambda: call('aod_conv2d_ws', C.byref(d), ptr(dz_rows), ptr(w_dgrad), ptr(out), None, None, ptr(res), ptr(mask),
                                   ptr(post_scale), None, ptr(colsum), ptr(ws), wsb, stream()), alg)
    return out


def _row_table(d, x_segs, dz_segs, Cin, Npad, R, S, stride, pad, dil, device):
    if len(_ROW_TABLES) > 512:
        _ROW_TABLES.clear()
    key = (tuple(x_segs), tuple(dz_segs), Cin, Npad, R, S, stride, pad, dil, device.index)
    tab = _ROW_TABLES.get(key)
    if tab is None:
        tab = torch.empty(max(int(_C.lib.aod_conv_row_table_bytes(C.byref(d))), 16), dtype=torch.uint8, device=device)
        call('aod_conv_row_table', C.byref(d), ptr(tab), stream())
        _ROW_TABLES[key] = tab
    return tab


class WgradJob:
    """One weight gradient waiting for its launch: operands of conv2d_wgrad_rows + destinations of unpack_wgrad (all preallocated)."""
    __slots__ = ('x_rows', 'x_segs', 'dz', 'dz_segs', 'R', 'S', 'stride', 'pad', 'dil', 'alg', 'O', 'I', 'gw', 'scale', 'w', 'wdot', 'bn',
                 'desc', 'scope')

    def __init__(self, x_rows, x_segs, dz, dz_segs, R, S, stride, pad, dil, alg, O, I, gw, scale=None, w=None, wdot=None, bn=None):
        self.x_rows, self.x_segs, self.dz, self.dz_segs = x_rows, x_segs, dz, dz_segs
        self.R, self.S, self.stride, self.pad, self.dil, self.alg, self.O, self.I = R, S, stride, pad, dil, alg, O, I
        self.gw, self.scale, self.w, self.wdot, self.bn = gw, scale, w, wdot, bn
        self.desc = make_desc(x_rows.shape[1], dz.shape[1], R, S, stride, pad, dil, x_segs, dz_segs, False, False, False)
        self.scope = SCOPE

    def run_alone(self):
        with scope(self.scope):
            self._run_alone()

    def _run_alone(self):
        dw = conv2d_wgrad_rows(self.x_rows, self.x_segs, self.dz, self.dz_segs, self.R, self.S, self.stride, self.pad, self.dil, alg=self.alg)
        unpack_wgrad(dw, self.O, self.I, grad_oihw=self.gw, scale=self.scale, w_oihw=self.w, want_wdot=self.wdot is not None, bn=self.bn,
                     wdot=self.wdot)


def wgrad_group_splits(jobs):
    """slab counts of the jobs as ONE grouped launch (aod_conv2d_wgrad_group_plan), or None when they cannot share a grid (mixed tile
    forms, more than 9 taps, slab form switched off)"""
    n = len(jobs)
    if n > 4 or not SLAB_WGRAD or any(j.R * j.S > 9 for j in jobs):
        return None
    splits = (C.c_int32 * n)()
    rc = _C.lib.aod_conv2d_wgrad_group_plan((C.c_void_p * n)(*[C.addressof(j.desc) for j in jobs]), n, splits)
    return list(splits) if rc == 0 else None


def wgrad_unpack_group(jobs):
    """weight gradients of up to four convs: one grouped wgrad launch + one grouped unpack (aod_conv2d_wgrad_grouped,
    aod_unpack_wgrad_slabs_grouped); a single job, or jobs that cannot share a grid, run as before"""
    splits = wgrad_group_splits(jobs) if len(jobs) > 1 else None
    if splits is None:
        for j in jobs:
            j.run_alone()
        return
    n = len(jobs)
    dev = jobs[0].dz.device
    # (x3: the launch writes LOGICAL slabs [N / 2][R][S][C / 2] -- the three head / tail bands of an entry are added in its epilogue)
    lg = 2 if X3 else 1
    strides = [(j.dz.shape[1] // lg) * j.R * j.S * (j.x_rows.shape[1] // lg) for j in jobs]
    offs, tot = [], 0
    for sp, st in zip(splits, strides):
        offs.append(tot)
        tot += sp * st
    buf = _slab_scratch(tot, dev)
    slabs = [buf[o:o + sp * st] for o, sp, st in zip(offs, splits, strides)]
    tabs = [_row_table(j.desc, j.x_segs, j.dz_segs, j.x_rows.shape[1], j.dz.shape[1], j.R, j.S, j.stride, j.pad, j.dil, dev) for j in jobs]
    PA, I32A, I64A = C.c_void_p * n, C.c_int32 * n, C.c_int64 * n
    pv = lambda ts: PA(*[(ptr(t).value if t is not None else None) for t in ts])

    def launch():
        call('aod_conv2d_wgrad_grouped', PA(*[C.addressof(j.desc) for j in jobs]), n, pv([j.x_rows for j in jobs]), pv([j.dz for j in jobs]),
             pv(slabs), I32A(*splits), I64A(*strides), pv(tabs), stream())
    ms = [sum(sg.B * sg.H * sg.W for sg in j.dz_segs) for j in jobs]
    global SCOPE
    SCOPE, prev_scope = jobs[0].scope, SCOPE            # (a group is listed under its first member's part of the network)
    flops = sum(2.0 * m * j.R * j.S * (j.alg[0] * j.alg[1] if j.alg is not None else j.dz.shape[1] * j.x_rows.shape[1]) for m, j in zip(ms, jobs))
    # (per-shape listings show the group as one line: rows of the first member, summed N x K)
    prof_flops('wgrad', (ms[0], sum(j.dz.shape[1] for j in jobs), sum(j.R * j.S * j.x_rows.shape[1] for j in jobs), 10 + n, 1), flops, launch)
    ws = [j.w.contiguous() if (j.wdot is not None and j.w is not None) else None for j in jobs]
    call('aod_unpack_wgrad_slabs_grouped', n, pv(slabs), I32A(*splits), I64A(*strides), pv([j.gw for j in jobs]), I32A(*[j.O for j in jobs]),
         I32A(*[j.I for j in jobs]), I32A(*[j.R for j in jobs]), I32A(*[j.S for j in jobs]), I32A(*[j.x_rows.shape[1] // lg for j in jobs]),
         I32A(*[0] * n), pv([j.scale for j in jobs]), pv(ws), pv([j.wdot for j in jobs]), pv([j.bn[0] if j.bn else None for j in jobs]),
         pv([j.bn[1] if j.bn else None for j in jobs]), pv([j.bn[2] if j.bn else None for j in jobs]), stream())
    SCOPE = prev_scope


def conv2d_wgrad_rows(x_rows, x_segs, dz_rows, dz_segs, R, S, stride=1, pad=0, dil=1, dw=None, alg=None):
    """dW of a conv: with `dw` given, [Npad][R][S][C] fp32 ACCUMULATED into it (fp32 atomics); otherwise the deterministic slab form --
    returns [nslabs][Npad][R][S][C] partial sums in a per-device scratch (valid until the next wgrad launch; unpack_wgrad adds them)."""
    Cin, Npad = x_rows.shape[1], dz_rows.shape[1]
    d = make_desc(Cin, Npad, R, S, stride, pad, dil, x_segs, dz_segs, False, False, False)
    nslabs = 0
    # (slab form: filters of at most 9 taps; in the reference-precision mode, which has no other form, up to 16 -- SSD512's 4 x 4 extra conv)
    if dw is None and R * S <= (16 if X3 else 9) and SLAB_WGRAD:
        nslabs = int(lib.aod_conv2d_wgrad_splits(C.byref(d)))
    assert not X3 or nslabs, 'x3 weight gradients exist in the slab form only (at most 16 taps)'
    if dw is None and nslabs == 0:
        dw = _dw_scratch(Npad * R * S * Cin, x_rows.device).view(Npad, R, S, Cin)
    tab = _row_table(d, x_segs, dz_segs, Cin, Npad, R, S, stride, pad, dil, x_rows.device)
    if nslabs:
        lg = 2 if X3 else 1                   # (x3: logical slabs, see wgrad_unpack_group)
        stride_ = (Npad // lg) * R * S * (Cin // lg)
        slabs = _slab_scratch(nslabs * stride_, x_rows.device).view(nslabs, Npad // lg, R, S, Cin // lg)
        _prof('wgrad', d, lambda: call('aod_conv2d_wgrad_slabs', C.byref(d), ptr(x_rows), ptr(dz_rows), ptr(slabs), nslabs, stride_, ptr(tab), stream()), alg)
        return slabs
    _prof('wgrad', d, lambda: call('aod_conv2d_wgrad', C.byref(d), ptr(x_rows), ptr(dz_rows), ptr(dw), ptr(tab), stream()), alg)
    return dw


_DW = {}
_SLABS = {}
SLAB_WGRAD = os.environ.get('AOD_WGRAD_SLABS', '1') != '0'      # debug switch: 0 = fp32-atomic accumulation into one dW (round-1 form)


def _slab_scratch(n, device):
    """Scratch for the wgrad slabs (at most 512 workgroups x 64 KB = 33.5 MB per launch), one per (device, stream): a launch and the unpack
    that reads its slabs are ordered by their stream, launches of two streams must not share one; need not be initialised.  A buffer first
    requested DURING a graph capture lives in that graph's private pool: it is keyed by the capture as well and never handed to eager code or
    to another capture; the cache is a small LRU -- streams come and go (capture side streams, the scoring pipeline)."""
    cap = torch.cuda.is_current_stream_capturing()
    key = (device, torch.cuda.current_stream(device).cuda_stream, _capture_tag() if cap else None)
    buf = _SLABS.pop(key, None)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, (1 << 23) + (1 << 21)), dtype=torch.float32, device=device)
    _SLABS[key] = buf                                   # (re-inserted: most recently used last)
    while len(_SLABS) > 12:
        # (graphs pin what their kernels captured -- graphs._pin_caches holds the dict's values at capture time -- so dropping the oldest
        # entry here cannot free memory a live graph points at)
        _SLABS.pop(next(iter(_SLABS)))
    return buf[:n]


_CAPTURE_SEQ = [0]


def _capture_tag():
    """changes from one graph capture to the next: reset_zero_arena() brackets every capture in graphs.py and bumps it"""
    return _CAPTURE_SEQ[0]


def _dw_scratch(n, device):
    """Persistent all-zero fp32 accumulator for wgrad (per device).  Invariant: all-zero between uses --
    `unpack_wgrad(clear=True)` zeroes what wgrad touched, so no memset launch is needed per conv."""
    buf = _DW.get(device)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 1 << 22), dtype=torch.float32, device=device)
        _DW[device] = buf
    return buf[:n]


_ZEROS = {}


def zeros_f32(n, device):
    """An all-zero fp32 vector carved out of a per-device arena (one fill per ~64K floats instead of one tiny fill launch per
    accumulator).  Slices are handed out once and never reused, so they may be kept as gradients."""
    n_al = (n + 63) // 64 * 64
    ar = _ZEROS.get(device)
    if ar is None or ar[1] + n_al > ar[0].numel():
        ar = [torch.zeros(max(1 << 16, n_al), dtype=torch.float32, device=device), 0]
        _ZEROS[device] = ar
    out = ar[0][ar[1]:ar[1] + n]
    ar[1] += n_al
    return out


def reset_zero_arena():
    """Drop the current arena: the next zeros_f32 call allocates (and fills) a fresh one.  Used around HIP-graph capture so that every
    accumulator used by captured kernels is zeroed by a captured fill.  (Also the boundary between captures for _slab_scratch.)"""
    _ZEROS.clear()
    _CAPTURE_SEQ[0] += 1


def unpack_wgrad(dw_orsi, O, I, grad_oihw=None, accumulate=False, clear=None, scale=None, w_oihw=None, want_wdot=False, bn=None, wdot=None):
    """Returns grad_oihw, or (grad_oihw, wdot) with wdot[o] = <w[o], dw[o]> when want_wdot; with bn = (s1, mean, invstd) wdot is the
    BatchNorm weight gradient invstd * (<w, dw> - mean * s1)."""
    if dw_orsi.dim() == 5:          # slab form (conv2d_wgrad_rows without `dw`)
        nslabs, Opad, R, S, Ipad = dw_orsi.shape
        if grad_oihw is None:
            grad_oihw = torch.empty(O, I, R, S, dtype=torch.float32, device=dw_orsi.device)
        if wdot is None:
            wdot = torch.empty(O, dtype=torch.float32, device=dw_orsi.device) if want_wdot else None
        call('aod_unpack_wgrad_slabs', ptr(dw_orsi), nslabs, Opad * R * S * Ipad, ptr(grad_oihw), O, I, R, S, Ipad, int(bool(accumulate)), ptr(scale),
             ptr(w_oihw.contiguous()) if want_wdot else None, ptr(wdot), ptr(bn[0]) if bn else None, ptr(bn[1]) if bn else None,
             ptr(bn[2]) if bn else None, stream())
        return (grad_oihw, wdot) if want_wdot else grad_oihw
    Opad, R, S, Ipad = dw_orsi.shape
    if grad_oihw is None:
        grad_oihw = torch.empty(O, I, R, S, dtype=torch.float32, device=dw_orsi.device)
    if clear is None:      # scratch-backed accumulators must go back all-zero; pad rows/channels only ever receive zeros
        buf = _DW.get(dw_orsi.device)
        clear = buf is not None and dw_orsi.data_ptr() == buf.data_ptr()
    if clear and (O != Opad or I != Ipad):
        # dZ pad columns / x pad channels are zero, so wgrad added exact zeros there: nothing to clear beyond [O, I]
        pass
    if wdot is None:
        wdot = torch.empty(O, dtype=torch.float32, device=dw_orsi.device) if want_wdot else None
    call('aod_unpack_wgrad', ptr(dw_orsi), ptr(grad_oihw), O, I, R, S, Ipad, int(accumulate), int(bool(clear)), ptr(scale),
         ptr(w_oihw.contiguous()) if want_wdot else None, ptr(wdot), ptr(bn[0]) if bn else None, ptr(bn[1]) if bn else None,
         ptr(bn[2]) if bn else None, stream())
    return (grad_oihw, wdot) if want_wdot else grad_oihw


def nchw_to_rows(img_f32, cpad=8):
    B, Cc, H, W = img_f32.shape
    out = torch.empty(B * H * W, cpad, dtype=torch.bfloat16, device=img_f32.device)
    call('aod_nchw_f32_to_nhwc_bf16', ptr(img_f32.contiguous()), ptr(out), B, Cc, H, W, cpad, stream())
    return out, [Seg(B, H, W, 0)]


def nchw_to_s2d_rows(img_f32):
    """fp32 [B, C <= 4, H, W] (H, W even) -> space-to-depth bf16 rows [B * H/2 * W/2, 16] (aod_nchw_f32_to_s2d_bf16)"""
    B, Cc, H, W = img_f32.shape
    out = torch.empty(B * (H // 2) * (W // 2), width(16), dtype=torch.bfloat16, device=img_f32.device)
    call('aod_x3_nchw_f32_to_s2d' if X3 else 'aod_nchw_f32_to_s2d_bf16', ptr(img_f32.contiguous()), ptr(out), B, Cc, H, W, stream())
    return out, [Seg(B, H // 2, W // 2, 0)]


def bottleneck64_fwd(x_rows, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, res_rows, out=None, ds=None):
    """aod_bottleneck64_fwd: y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + res) for a 64-channel bottleneck, one launch.
    ds = (wd, sd, bd) (reference-precision mode, 64 input channels): res = bn_d(conv_d(x)) computed inside the launch (aod_bottleneck64x3_ds_fwd)"""
    M, Cin = x_rows.shape
    if X3 and ds is not None:
        assert M == B * H * W and Cin == 128 and res_rows is None
        if out is None:
            out = torch.empty(M, 512, dtype=torch.bfloat16, device=x_rows.device)
        prof_flops('fwd', (M, 256, 2 * (64 + 576 + 64 + 64), 12, 1), 2.0 * M * (64 * 64 + 576 * 64 + 64 * 256 + 64 * 256),
                   lambda: call('aod_bottleneck64x3_ds_fwd', ptr(x_rows), B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3),
                                ptr(b3), ptr(ds[0]), ptr(ds[1]), ptr(ds[2]), ptr(out), stream()))
        return out
    if X3:          # X rows: the x3 twin of the kernel (csrc/bottleneck_x3.hip); Cin = logical input channels
        Cl = Cin // 2
        assert M == B * H * W and res_rows.shape == (M, 512)
        if out is None:
            out = torch.empty(M, 512, dtype=torch.bfloat16, device=x_rows.device)
        prof_flops('fwd', (M, 256, 2 * (Cl + 576 + 64), 11, 1), 2.0 * M * (Cl * 64 + 576 * 64 + 64 * 256),
                   lambda: call('aod_bottleneck64x3_fwd', ptr(x_rows), Cl, B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3),
                                ptr(b3), ptr(res_rows), ptr(out), stream()))
        return out
    if ds is not None:          # (fast mode: the same, aod_bottleneck64_ds_fwd)
        assert M == B * H * W and Cin == 64 and res_rows is None
        if out is None:
            out = torch.empty(M, 256, dtype=torch.bfloat16, device=x_rows.device)
        prof_flops('fwd', (M, 256, 64 + 576 + 64 + 64, 12, 1), 2.0 * M * (64 * 64 + 576 * 64 + 64 * 256 + 64 * 256),
                   lambda: call('aod_bottleneck64_ds_fwd', ptr(x_rows), B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3),
                                ptr(b3), ptr(ds[0]), ptr(ds[1]), ptr(ds[2]), ptr(out), stream()))
        return out
    assert M == B * H * W and res_rows.shape == (M, 256)
    if out is None:
        out = torch.empty(M, 256, dtype=torch.bfloat16, device=x_rows.device)
    flops = 2.0 * M * (Cin * 64 + 576 * 64 + 64 * 256)
    prof_flops('fwd', (M, 256, Cin + 576 + 64, 11, 1), flops,
               lambda: call('aod_bottleneck64_fwd', ptr(x_rows), Cin, B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3),
                            ptr(b3), ptr(res_rows), ptr(out), stream()))
    return out


def bottleneck128_fwd(x_rows, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, keep=False, frag=False):
    """aod_bottleneck128_fwd / aod_bottleneck256_fwd (by the channel count of x: 512 / 1024): identity bottleneck of the 128- / 256-plane
    stage in one launch; keep=True also returns the intermediates t1, t2"""
    M, Cin = x_rows.shape
    if X3:          # X rows: the x3 twin of the 128-plane kernel (csrc/bottleneck128_x3.hip)
        assert M == B * H * W and Cin == 1024 and not frag
        out = torch.empty(M, 1024, dtype=torch.bfloat16, device=x_rows.device)
        t1 = torch.empty(M, 256, dtype=torch.bfloat16, device=x_rows.device) if keep else None
        t2 = torch.empty(M, 256, dtype=torch.bfloat16, device=x_rows.device) if keep else None
        prof_flops('fwd', (M, 512, 2 * (512 + 9 * 128 + 128), 11, 1), 2.0 * M * (512 * 128 + 9 * 128 * 128 + 128 * 512),
                   lambda: call('aod_bottleneck128x3_fwd', ptr(x_rows), B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3),
                                ptr(b3), ptr(out), ptr(t1), ptr(t2), stream()))
        return (out, t1, t2) if keep else out
    assert M == B * H * W and Cin in (512, 1024)
    Pl = Cin // 4
    out = torch.empty(M, Cin, dtype=torch.bfloat16, device=x_rows.device)
    t1 = torch.empty(M, Pl, dtype=torch.bfloat16, device=x_rows.device) if keep else None
    t2 = torch.empty(M, Pl, dtype=torch.bfloat16, device=x_rows.device) if keep else None
    flops = 2.0 * M * (Cin * Pl + 9 * Pl * Pl + Pl * Cin)
    prof_flops('fwd', (M, Cin, Cin + 9 * Pl + Pl, 11, 1), flops,
               lambda: call('aod_bottleneck128_fwd' if Pl == 128 else ('aod_bottleneck256f_fwd' if frag else 'aod_bottleneck256_fwd'), ptr(x_rows), B, H, W, ptr(w1), ptr(s1), ptr(b1), ptr(w2), ptr(s2), ptr(b2), ptr(w3), ptr(s3), ptr(b3),
                            ptr(out), ptr(t1), ptr(t2), stream()))
    return (out, t1, t2) if keep else out


def bottleneck_bwd(g_rows, B, H, W, wd3, wd2, wd1, act_t2, act_t1, act_x, frag=False):
    """aod_bottleneck_bwd: dgrad chain of an identity bottleneck (128 / 256 planes) in one launch -> (gx, gt2, gt1, colsum_x, colsum_t2, colsum_t1)"""
    M, Cin = g_rows.shape
    if X3:          # X rows, 128 planes (csrc/bottleneck128_x3.hip, the BWD instance)
        assert M == B * H * W and Cin == 1024 and act_x.shape == g_rows.shape and act_t2.shape == (M, 256) and act_t1.shape == (M, 256) and not frag
        dev = g_rows.device
        gx = torch.empty(M, 1024, dtype=torch.bfloat16, device=dev)
        gt2 = torch.empty(M, 256, dtype=torch.bfloat16, device=dev)
        gt1 = torch.empty(M, 256, dtype=torch.bfloat16, device=dev)
        cx, c2, c1 = zeros_f32(512, dev), zeros_f32(128, dev), zeros_f32(128, dev)
        prof_flops('dgrad', (M, 512, 2 * (512 + 9 * 128 + 128), 11, 1), 2.0 * M * (512 * 128 + 9 * 128 * 128 + 128 * 512),
                   lambda: call('aod_bottleneck128x3_bwd', ptr(g_rows), B, H, W, ptr(wd3), ptr(wd2), ptr(wd1), ptr(act_t2), ptr(act_t1), ptr(act_x),
                                ptr(gx), ptr(gt2), ptr(gt1), ptr(c2), ptr(c1), ptr(cx), stream()))
        return gx, gt2, gt1, cx, c2, c1
    assert M == B * H * W and Cin in (512, 1024) and act_x.shape == g_rows.shape
    Pl = Cin // 4
    assert act_t2.shape == (M, Pl) and act_t1.shape == (M, Pl)
    dev = g_rows.device
    gx = torch.empty(M, Cin, dtype=torch.bfloat16, device=dev)
    gt2 = torch.empty(M, Pl, dtype=torch.bfloat16, device=dev)
    gt1 = torch.empty(M, Pl, dtype=torch.bfloat16, device=dev)
    cx, c2, c1 = zeros_f32(Cin, dev), zeros_f32(Pl, dev), zeros_f32(Pl, dev)
    flops = 2.0 * M * (Cin * Pl + 9 * Pl * Pl + Pl * Cin)
    prof_flops('dgrad', (M, Cin, Cin + 9 * Pl + Pl, 11, 1), flops,
               lambda: (call('aod_bottleneck256f_bwd', ptr(g_rows), B, H, W, ptr(wd3), ptr(wd2), ptr(wd1), ptr(act_t2), ptr(act_t1), ptr(act_x),
                             ptr(gx), ptr(gt2), ptr(gt1), ptr(c2), ptr(c1), ptr(cx), stream()) if frag and Pl == 256 else
                        call('aod_bottleneck_bwd', Pl, ptr(g_rows), B, H, W, ptr(wd3), ptr(wd2), ptr(wd1), ptr(act_t2), ptr(act_t1), ptr(act_x),
                             ptr(gx), ptr(gt2), ptr(gt1), ptr(c2), ptr(c1), ptr(cx), stream())))
    return gx, gt2, gt1, cx, c2, c1


def rows_to_nchw(rows, seg: Seg):
    """View rows of one segment as a [B, C, H, W] channels_last tensor (no copy)."""
    Cc = rows.shape[1]
    return rows[seg.row0:seg.row0 + seg.rows].view(seg.B, seg.H, seg.W, Cc).permute(0, 3, 1, 2)


def maxpool3x3s2(x_rows, seg: Seg):
    Cc = x_rows.shape[1]
    oh, ow = out_hw(seg.H, seg.W, 3, 3, 2, 1, 1)
    out = torch.empty(seg.B * oh * ow, Cc, dtype=torch.bfloat16, device=x_rows.device)
    call('aod_x3_maxpool3x3s2' if X3 else 'aod_maxpool3x3s2', ptr(x_rows), ptr(out), seg.B, seg.H, seg.W, Cc, stream())
    return out, Seg(seg.B, oh, ow, 0)


def upsample_add_(dst_rows, dst_seg: Seg, src_rows, src_seg: Seg):
    call('aod_upsample2x_add', ptr(src_rows), ptr(dst_rows), dst_seg.B, src_seg.H, src_seg.W, dst_rows.shape[1],
         dst_seg.H, dst_seg.W, stream())
    return dst_rows


def upsample_add(lat_rows, dst_seg: Seg, top_rows, src_seg: Seg):
    """out = lateral + nearest_upsample(top), out of place (aod_upsample2x_add_to)"""
    out = torch.empty_like(lat_rows)
    call('aod_x3_upsample2x_add_to' if X3 else 'aod_upsample2x_add_to', ptr(top_rows), ptr(lat_rows), ptr(out), dst_seg.B, src_seg.H, src_seg.W, lat_rows.shape[1], dst_seg.H, dst_seg.W, stream())
    return out


def upsample_add_bwd(g_dst_rows, dst_seg: Seg, src_seg: Seg):
    """g_top = adjoint of the nearest upsample applied to g (written, not accumulated: no zero fill)"""
    out = torch.empty(src_seg.rows, g_dst_rows.shape[1], dtype=torch.bfloat16, device=g_dst_rows.device)
    call('aod_x3_upsample2x_add_bwd_set' if X3 else 'aod_upsample2x_add_bwd_set', ptr(g_dst_rows), ptr(out), dst_seg.B, src_seg.H, src_seg.W, g_dst_rows.shape[1], dst_seg.H, dst_seg.W, stream())
    return out


def upsample_add_bwd_(g_src_rows, src_seg: Seg, g_dst_rows, dst_seg: Seg):
    call('aod_upsample2x_add_bwd', ptr(g_dst_rows), ptr(g_src_rows), dst_seg.B, src_seg.H, src_seg.W,
         g_dst_rows.shape[1], dst_seg.H, dst_seg.W, stream())
    return g_src_rows


def add_relu(a, b):
    out = torch.empty_like(a)
    call('aod_add_relu', ptr(a), ptr(b), ptr(out), a.numel(), stream())
    return out


def act_bwd(g, a=None, z=None, scale=None, mean=None, invstd=None, relu=True, want_gm=False, want_dz=True):
    """Returns (dz, gm, dbeta, dgamma).  g [M, N] bf16 or fp32."""
    M, N = g.shape
    if X3:          # X rows in, X rows out; column sums over the N / 2 logical channels (aod_x3_act_bwd)
        assert z is None and scale is None and not want_gm and g.dtype == torch.bfloat16
        dz = torch.empty(M, N, dtype=torch.bfloat16, device=g.device) if want_dz else None
        dbeta = zeros_f32(N // 2, g.device)
        call('aod_x3_act_bwd', ptr(g), ptr(a), ptr(dz), ptr(dbeta), M, N, int(relu), stream())
        return dz, None, dbeta, None
    dz = torch.empty(M, N, dtype=torch.bfloat16, device=g.device) if want_dz else None
    gm = torch.empty(M, N, dtype=torch.bfloat16, device=g.device) if want_gm else None
    dbeta = zeros_f32(N, g.device)
    dgamma = zeros_f32(N, g.device) if z is not None else None
    call('aod_act_bwd', ptr(g), ptr(a), ptr(z), ptr(scale), ptr(mean), ptr(invstd), ptr(dz), ptr(gm), ptr(dbeta),
         ptr(dgamma), M, N, int(relu), int(g.dtype == torch.float32), stream())
    return dz, gm, dbeta, dgamma


def edl_focal_l1_fwd(cls, labels, label_w, bbox_pred=None, bbox_tgt=None, bbox_w=None, gamma=2.0, alpha=0.25, sums=None):
    """cls [rows, C] fp32.  Returns (loss_noR [rows], sums[3] = (sum l*w, sum |d|*bw, sum noR))."""
    rows, Cc = cls.shape
    loss_noR = torch.empty(rows, dtype=torch.float32, device=cls.device)
    if sums is None:
        sums = zeros_f32(3, cls.device)
    part = torch.empty(max(int(_C.lib.aod_loss_partials_len(rows)), 1), dtype=torch.float32, device=cls.device)
    # algorithmic bytes per anchor row: logits + label (int64) + label weight + 3 box vectors in, loss_noR out
    prof_bytes('edl_l1_fwd', rows * (Cc * 4 + 8 + 4 + (48 if bbox_pred is not None else 0) + 4),
               lambda: call('aod_edl_focal_l1_fwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox_pred), ptr(bbox_tgt), ptr(bbox_w), rows, Cc,
                            gamma, alpha, ptr(loss_noR), ptr(sums), ptr(part), stream()))
    return loss_noR, sums


def edl_focal_l1_bwd(cls, labels, label_w, bbox_pred, bbox_tgt, bbox_w, g_cls, g_bbox, g_noR=None, g_noR_scalar=0.0,
                     gamma=2.0, alpha=0.25, g_noR_is_scalar=False, out_bf16=False, A=1, pitch_cls=None, pitch_box=None, grad_cls=None, grad_bbox=None):
    rows, Cc = cls.shape
    pitch_cls = pitch_cls or A * Cc
    pitch_box = pitch_box or A * 4
    dt = torch.bfloat16 if out_bf16 else torch.float32
    # (the kernel writes every element of an unpadded row: only pad columns need the zero fill)
    alloc = lambda pitch, width: (torch.empty if pitch == width else torch.zeros)(rows // A, pitch, dtype=dt, device=cls.device)
    if grad_cls is None:
        grad_cls = alloc(pitch_cls, A * Cc)
    if grad_bbox is None and bbox_pred is not None:
        grad_bbox = alloc(pitch_box, A * 4)
    esz = 2 if out_bf16 else 4
    prof_bytes('edl_l1_bwd', rows * (Cc * 4 + 8 + 4 + (4 if (g_noR is not None and not g_noR_is_scalar) else 0) + Cc * esz + ((48 + 4 * esz) if bbox_pred is not None else 0)),
               lambda: call('aod_edl_focal_l1_bwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox_pred), ptr(bbox_tgt), ptr(bbox_w), rows, Cc,
                            gamma, alpha, ptr(g_cls), ptr(g_bbox), ptr(g_noR), float(g_noR_scalar), int(bool(g_noR_is_scalar)), ptr(grad_cls), ptr(grad_bbox), int(out_bf16), A,
                            pitch_cls, pitch_box, stream()))
    return grad_cls, grad_bbox


def meh_loss_fwd(lam, loss_noR, bbox_w4, out_sum=None):
    n = lam.numel()
    if out_sum is None:
        out_sum = zeros_f32(1, lam.device)
    part = torch.empty(max(int(_C.lib.aod_loss_partials_len(n)), 1), dtype=torch.float32, device=lam.device)
    call('aod_meh_loss_fwd', ptr(lam), ptr(loss_noR), ptr(bbox_w4), n, ptr(out_sum), ptr(part), stream())
    return out_sum


def meh_loss_bwd(lam, loss_noR, bbox_w4, g, out_bf16=False, A=1, pitch=None, grad=None):
    n = lam.numel()
    pitch = pitch or A
    if grad is None:
        grad = (torch.empty if pitch == A else torch.zeros)(n // A, pitch, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=lam.device)
    call('aod_meh_loss_bwd', ptr(lam), ptr(loss_noR), ptr(bbox_w4), n, ptr(g), ptr(grad), int(out_bf16), A, pitch, stream())
    return grad


def _level_rows(rows):
    """HOST int64 array of per-level row counts for the level-fused loss launches (kept alive by the caller for the duration of the call)."""
    return (C.c_int64 * len(rows))(*[int(r) for r in rows])


def edl_focal_l1_levels_fwd(cls, labels, label_w, bbox_pred, bbox_tgt, bbox_w, level_rows, gamma=2.0, alpha=0.25, num_pos=None):
    """All pyramid levels in one launch: cls [rows, C] fp32 with level l = rows [sum(level_rows[:l]), +level_rows[l]).  Returns
    (loss_noR [rows], sums [3, L]) -- bit-identical to L calls of edl_focal_l1_fwd on the level slices.  With num_pos (int32 [B], the
    assigner's per-image positive counts): (loss_noR, sums / divisors, divisors [3, L], num_total_samples [1])."""
    rows, Cc = cls.shape
    L = len(level_rows)
    assert sum(level_rows) == rows
    lr = _level_rows(level_rows)
    loss_noR = torch.empty(rows, dtype=torch.float32, device=cls.device)
    sums = torch.empty(3, L, dtype=torch.float32, device=cls.device)
    div = nt = None
    if num_pos is not None:
        assert num_pos.dtype == torch.int32 and num_pos.is_contiguous()
        div = torch.empty(3, L, dtype=torch.float32, device=cls.device)
        nt = torch.empty(1, dtype=torch.float32, device=cls.device)
    part = torch.empty(max(int(_C.lib.aod_loss_levels_partials_len(L, lr)), 1), dtype=torch.float32, device=cls.device)
    prof_bytes('edl_l1_fwd', rows * (Cc * 4 + 8 + 4 + (48 if bbox_pred is not None else 0) + 4),
               lambda: call('aod_edl_focal_l1_levels_fwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox_pred), ptr(bbox_tgt), ptr(bbox_w), L, lr, Cc,
                            gamma, alpha, ptr(loss_noR), ptr(sums), ptr(part), ptr(num_pos), 0 if num_pos is None else int(num_pos.numel()), ptr(div),
                            ptr(nt), stream()))
    return (loss_noR, sums) if num_pos is None else (loss_noR, sums, div, nt)


def edl_focal_l1_levels_bwd(cls, labels, label_w, bbox_pred, bbox_tgt, bbox_w, level_rows, g_sums, g_noR_rows, grad_cls, grad_bbox, A,
                            gamma=2.0, alpha=0.25, divisors=None):
    """g_sums [3, L] fp32 contiguous; gradients are written into grad_cls [rows / A, A * C] / grad_bbox [rows / A, A * 4] (fp32, unpadded)."""
    rows, Cc = cls.shape
    L = len(level_rows)
    lr = _level_rows(level_rows)
    assert g_sums.shape == (3, L) and g_sums.is_contiguous() and g_sums.dtype == torch.float32
    prof_bytes('edl_l1_bwd', rows * (Cc * 4 + 8 + 4 + (4 if g_noR_rows is not None else 0) + Cc * 4 + ((48 + 16) if bbox_pred is not None else 0)),
               lambda: call('aod_edl_focal_l1_levels_bwd', ptr(cls), ptr(labels), ptr(label_w), ptr(bbox_pred), ptr(bbox_tgt), ptr(bbox_w), L, lr, Cc,
                            gamma, alpha, ptr(g_sums), ptr(divisors), ptr(g_noR_rows), ptr(grad_cls), ptr(grad_bbox), 0, A, A * Cc, A * 4, stream()))
    return grad_cls, grad_bbox


def meh_loss_levels_fwd(lam, loss_noR, bbox_w4, level_rows):
    L = len(level_rows)
    assert sum(level_rows) == lam.numel()
    lr = _level_rows(level_rows)
    out = torch.empty(L, dtype=torch.float32, device=lam.device)
    part = torch.empty(max(int(_C.lib.aod_loss_levels_partials_len(L, lr)), 1), dtype=torch.float32, device=lam.device)
    call('aod_meh_loss_levels_fwd', ptr(lam), ptr(loss_noR), ptr(bbox_w4), L, lr, ptr(out), ptr(part), stream())
    return out


def meh_loss_levels_bwd(lam, loss_noR, bbox_w4, level_rows, g, grad, A):
    L = len(level_rows)
    lr = _level_rows(level_rows)
    assert g.shape == (L,) and g.is_contiguous() and g.dtype == torch.float32
    call('aod_meh_loss_levels_bwd', ptr(lam), ptr(loss_noR), ptr(bbox_w4), L, lr, ptr(g), ptr(grad), 0, A, A, stream())
    return grad


def pad_cast_colsum(g, npad, relu_out=None):
    M, N = g.shape
    assert relu_out is None or relu_out.dtype == torch.float32
    if X3:          # fp32 head gradients -> X rows of xw(N) columns (npad is that width)
        assert g.dtype == torch.float32 and npad == xw(N)
        dz = torch.empty(M, npad, dtype=torch.bfloat16, device=g.device)
        cs = zeros_f32(npad // 2, g.device)
        call('aod_x3_pad_cast_colsum', ptr(g), ptr(relu_out), ptr(dz), ptr(cs), M, N, stream())
        return dz, cs
    dz = torch.empty(M, npad, dtype=torch.bfloat16, device=g.device)
    cs = zeros_f32(npad, g.device)
    call('aod_pad_cast_colsum', ptr(g), ptr(relu_out), ptr(dz), ptr(cs), M, N, npad, int(g.dtype == torch.float32), stream())
    return dz, cs


_F4 = C.c_float * 4


def max_iou_assign(anchors, valid, gts, gt_count, gt_labels, pos_thr=0.5, neg_thr=0.4, min_pos_iou=0.0, assign_all=True,
                   num_classes=20, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), level_start=None):
    """anchors [A,4] f32, valid [B,A] uint8/bool or None, gts [B,Gmax,4], gt_count [B] int32, gt_labels [B,Gmax] int64."""
    A = anchors.shape[0]
    B, Gmax = gts.shape[:2]
    dev = anchors.device
    assigned = torch.empty(B, A, dtype=torch.int64, device=dev)
    labels = torch.empty(B, A, dtype=torch.int64, device=dev)
    label_w = torch.empty(B, A, dtype=torch.float32, device=dev)
    bbox_t = torch.empty(B, A, 4, dtype=torch.float32, device=dev)
    bbox_w = torch.empty(B, A, 4, dtype=torch.float32, device=dev)
    num_pos = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(_C.lib.aod_assign_ws_bytes(B, Gmax)), 8), dtype=torch.uint8, device=dev)
    if valid is not None:
        valid = valid.view(torch.uint8) if valid.dtype == torch.bool else valid
    call('aod_max_iou_assign', ptr(anchors), ptr(valid), A, B, ptr(gts), ptr(gt_count), ptr(gt_labels), Gmax, pos_thr, neg_thr,
         min_pos_iou, int(assign_all), num_classes, _F4(*means), _F4(*stds), ptr(assigned), ptr(labels), ptr(label_w), ptr(bbox_t),
         ptr(bbox_w), ptr(num_pos), ptr(ws), 0 if level_start is None else len(level_start) - 1,
         None if level_start is None else (C.c_int64 * len(level_start))(*level_start), stream())
    return assigned, labels, label_w, bbox_t, bbox_w, num_pos


def x3_split(t_f32):
    """fp32 [M, C] -> X rows [M, xw(C)] (aod_x3_split)"""
    M, Cc = t_f32.shape
    out = torch.empty(M, xw(Cc), dtype=torch.bfloat16, device=t_f32.device)
    call('aod_x3_split', ptr(t_f32.contiguous()), ptr(out), M, Cc, stream())
    return out


def x3_merge(x_rows, Cc=None):
    """X rows [M, 2 * Cp] -> fp32 [M, C] (C defaults to the padded count Cp)"""
    M, Wd = x_rows.shape
    Cc = Cc or Wd // 2
    out = torch.empty(M, Cc, dtype=torch.float32, device=x_rows.device)
    call('aod_x3_merge', ptr(x_rows.contiguous()), ptr(out), M, Cc, stream())
    return out


def x3_add(a, b):
    out = torch.empty_like(a)
    call('aod_x3_add', ptr(a), ptr(b), ptr(out), a.numel(), stream())
    return out
