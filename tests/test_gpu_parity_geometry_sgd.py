"""GPU parity of the integer-exact rows through the C ABI: anchors / valid flags on the device, MaxIoU assignment + pseudo
sampling + delta encoding (a6) against the REFERENCE's golden arrays (tests/golden/assign.npz, anchors.npz) and the oracle, and the fused
multi-tensor SGD (a11) against torch.optim.SGD and oracle.model.sgd_step on identical gradients."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import geometry as ogeo
from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
STR = (8, 16, 32, 64, 128)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def _ag():
    from aod_meh_hua_amd.core.anchor import AnchorGenerator
    return AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=list(STR))


# ------------------------------------------------------------------ a5: anchors / valid flags as they sit in HBM
def test_device_anchors_and_valid_flags_bit_exact_vs_reference():
    """anchor_generator.py:308-438: the tensors the assign / decode kernels actually read on the device."""
    g = np.load(os.path.join(G, 'anchors.npz'))
    ag = _ag()
    sizes512 = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    big = ag.grid_anchors(sizes512, 'cuda')
    assert all(a.is_cuda and a.dtype == torch.float32 for a in big)
    assert [sha(a) for a in big] == list(g['grid512_sha'])
    assert np.array_equal(torch.stack([a[:9] for a in big]).cpu().numpy(), g['grid512_first'])
    assert np.array_equal(torch.stack([a[-9:] for a in big]).cpu().numpy(), g['grid512_last'])
    flat = ag.flat_grid_anchors(sizes512, 'cuda')
    assert flat.shape == (49104, 4) and torch.equal(flat, torch.cat(big))
    small_sizes = [(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)]
    small = ag.grid_anchors(small_sizes, 'cuda')
    assert np.array_equal(torch.cat(small).cpu().numpy(), g['grid_small'])
    f_small = ag.valid_flags(small_sizes, (120, 96, 3), 'cuda')
    assert all(f.is_cuda for f in f_small)
    assert np.array_equal(torch.cat(f_small).cpu().numpy().astype(bool), g['flags_small'])
    fb = ag.valid_flags(sizes512, (480, 500, 3), 'cuda')
    assert [sha(x.bool()) for x in fb] == list(g['flags512b_sha'])
    assert [int(x.sum()) for x in fb] == list(g['flags512b_sum'])
    fa = ag.valid_flags(sizes512, (512, 512, 3), 'cuda')
    assert [int(x.sum()) for x in fa] == list(g['flags512a_sum'])


# ------------------------------------------------------------------ a6: assignment kernel
def _pack(gtb, gtl, dev='cuda'):
    from aod_meh_hua_amd.models.dense_heads.L_anchor_head import pack_gts
    return pack_gts(gtb, gtl, dev)


def _assign(flat, valid, gtb, gtl, level_start=None, **kw):
    from aod_meh_hua_amd import hipops as ho
    gts, counts, labs = _pack(gtb, gtl)
    out = ho.max_iou_assign(flat, valid, gts, counts, labs, level_start=level_start, **kw)
    torch.cuda.synchronize()
    return out


def test_assign_kernel_vs_reference_golden_cases():
    """max_iou_assigner.py:127-210 on the three golden images: G = 0 (:145-161), a gt whose best IoU < 0.5 (kept through the gt-max
    pass, min_pos_iou = 0), duplicate gts (argmax keeps the first, the gt-max pass hands ties to the LATER gt, :193-199), with both
    assigner configurations of the repo (RetinaNet 0.5/0.4/assign_all; SSD 0.5/0.5/single argmax)."""
    g = np.load(os.path.join(G, 'assign.npz'))
    ag = _ag()
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    mlvl = ag.grid_anchors(sizes, 'cuda')
    flat = ag.flat_grid_anchors(sizes, 'cuda')
    gtb, gtl = synth.assign_cases(128, 128)
    assigned, labels, lw, bt, bw, num_pos = _assign(flat, None, gtb, gtl)
    assert assigned.dtype == torch.int64 and labels.dtype == torch.int64
    assert np.array_equal(assigned.cpu().numpy(), g['gt_inds'])
    assert np.array_equal(labels.cpu().numpy(), g['labels'])
    assert np.array_equal(lw.cpu().numpy(), g['label_weights'])
    assert np.array_equal(bw.cpu().numpy(), g['bbox_weights'])
    got_bt, exp_bt = bt.cpu().numpy(), g['bbox_targets']
    assert np.allclose(got_bt, exp_bt, rtol=0, atol=2e-6), np.abs(got_bt - exp_bt).max()      # logf ulps only
    assert np.array_equal(got_bt == 0, exp_bt == 0)
    assert int(num_pos.clamp(min=1).sum()) == int(g['num_total_pos'])
    assert int((lw > 0).sum() - (bw[..., 0] > 0).sum()) == int(g['num_total_neg'])
    # SSD assigner configuration (Config_SSD.py:56-62): neg_iou_thr = 0.5, gt_max_assign_all = False
    a2, *_ = _assign(flat, None, gtb, gtl, pos_thr=0.5, neg_thr=0.5, min_pos_iou=0.0, assign_all=False)
    assert np.array_equal(a2.cpu().numpy(), g['gt_inds_ssdcfg'])
    # level-major output form (images_to_levels without a copy) carries the same values
    starts = [0]
    for a in mlvl:
        starts.append(starts[-1] + a.shape[0])
    al, ll, lwl, btl, bwl, npl = _assign(flat, None, gtb, gtl, level_start=starts)
    B = 3
    for name, lm, ref in (('assigned', al, assigned), ('labels', ll, labels), ('lw', lwl, lw)):
        for l in range(5):
            blk = lm.view(-1)[starts[l] * B:starts[l + 1] * B].view(B, -1)
            assert torch.equal(blk, ref[:, starts[l]:starts[l + 1]]), (name, l)
    for lm, ref in ((btl, bt), (bwl, bw)):
        for l in range(5):
            blk = lm.view(-1, 4)[starts[l] * B:starts[l + 1] * B].view(B, -1, 4)
            assert torch.equal(blk, ref[:, starts[l]:starts[l + 1]])
    assert torch.equal(npl, num_pos)


@pytest.mark.parametrize('cfg', ['retina', 'ssd'])
def test_assign_kernel_vs_oracle_random_with_ties_and_invalid_anchors(cfg):
    """Randomised images at a non-square padded shape (so some anchors are invalid: anchor/utils.py:20-46 with allowed_border = -1),
    G from 0 to 40 with exact duplicates and anchor-aligned boxes (tie rules), every output compared with the oracle."""
    ag = _ag()
    H, W = 224, 160
    sizes = [(28, 20), (14, 10), (7, 5), (4, 3), (2, 2)]
    pad_shapes = [(H, W, 3), (200, 150, 3), (H, W, 3), (97, 160, 3), (H, W, 3), (H, W, 3)]
    flat = ag.flat_grid_anchors(sizes, 'cuda')
    flags = torch.stack([ag.flat_valid_flags(sizes, ps[:2], 'cuda') for ps in pad_shapes])
    assert not bool(flags.all()) and flags.dtype in (torch.bool, torch.uint8)
    gen = torch.Generator().manual_seed(1234)
    gtb, gtl = [], []
    anchors_cpu = flat.cpu()
    for b, G_ in enumerate((0, 1, 7, 40, 3, 12)):
        wh = torch.rand(G_, 2, generator=gen) * torch.tensor([W * 0.7, H * 0.7]) + 4.0
        xy = torch.rand(G_, 2, generator=gen) * (torch.tensor([float(W), float(H)]) - wh).clamp(min=0)
        bx = torch.cat([xy, xy + wh], 1)
        if G_ >= 3:
            bx[-1] = bx[0]                                                   # exact duplicate gt
            ok = ((anchors_cpu[:, 0] >= 0) & (anchors_cpu[:, 1] >= 0) & (anchors_cpu[:, 2] <= W) & (anchors_cpu[:, 3] <= H)).nonzero()[:, 0]
            bx[1] = anchors_cpu[ok[int(torch.randint(0, len(ok), (1,), generator=gen))]]      # anchor-aligned gt (IoU exactly 1, ties)
        gtb.append(bx)
        gtl.append(torch.randint(0, 20, (G_,), generator=gen))
    kw = dict(retina=dict(), ssd=dict(pos_thr=0.5, neg_thr=0.5, min_pos_iou=0.0, assign_all=False, stds=(.1, .1, .2, .2)))[cfg]
    assigned, labels, lw, bt, bw, num_pos = _assign(flat, flags, gtb, gtl, **kw)
    mlvl = [a.cpu() for a in ag.grid_anchors(sizes, 'cuda')]
    ofl = [ogeo.valid_flags(sizes, STR, ps, [9] * 5) for ps in pad_shapes]
    ocfg = None if cfg == 'retina' else dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.0, gt_max_assign_all=False)
    tg = ogeo.get_targets(mlvl, ofl, gtb, gtl, 20, ocfg, coder_stds=(1., 1., 1., 1.) if cfg == 'retina' else (.1, .1, .2, .2))
    assert torch.equal(torch.cat([torch.cat(f) for f in ofl]).view(len(pad_shapes), -1), flags.cpu().bool())
    # invalid anchors: the reference never assigns them (unmap fill): gt_inds -1 is the oracle's marker for "not inside"
    exp_inds = tg['assigned_gt_inds']
    got = assigned.cpu()
    inside = flags.cpu().bool()
    assert torch.equal(got[inside], exp_inds[inside])
    assert torch.equal(labels.cpu(), torch.cat(tg['labels'], 1))
    assert torch.equal(lw.cpu(), torch.cat(tg['label_weights'], 1))
    assert torch.equal(bw.cpu(), torch.cat(tg['bbox_weights'], 1))
    ebt = torch.cat(tg['bbox_targets'], 1)
    assert torch.allclose(bt.cpu(), ebt, rtol=0, atol=2e-5 if cfg == 'ssd' else 2e-6), float((bt.cpu() - ebt).abs().max())
    assert int(num_pos.clamp(min=1).sum()) == tg['num_total_pos']
    assert int((got > 0).sum()) > 50 and bool((got[0][inside[0]] == 0).all())                # image 0 has no gt: all background


# ------------------------------------------------------------------ a11: fused SGD
def _param_set(seed=0):
    gen = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (256,), (256, 64, 1, 1), (1,), (9, 256, 3, 3), (180,), (512, 128, 3, 3), (1000003,), (7, 5)]
    return [torch.randn(*s, generator=gen) * 0.05 for s in shapes]


def test_fused_sgd_matches_torch_sgd_and_oracle_over_steps_and_lr_changes():
    """apis/train_Lambda.py:54-61: SGD(lr, momentum 0.9, wd 1e-4) for the main parameters and a second one for the MEH parameters; the
    LR schedule only touches the first (StepLrUpdaterHook).  Three steps on identical gradients, LR decayed before step 3, with the
    device-resident LR (HIP-graph mode) switched on after step 1."""
    from aod_meh_hua_amd.optim import FusedSGD
    p0 = _param_set()
    main_idx, meh_idx = [0, 1, 2, 3, 6, 7, 8], [4, 5]
    fused = [torch.nn.Parameter(p.clone().cuda()) for p in p0]
    ref = [torch.nn.Parameter(p.clone().double()) for p in p0]                        # fp64 torch.optim.SGD = the ideal
    ref32 = [torch.nn.Parameter(p.clone()) for p in p0]                              # fp32 torch.optim.SGD (what the reference runs)
    orc = {str(i): p.clone() for i, p in enumerate(p0)}
    bufs_main, bufs_L = {}, {}
    kw = dict(lr=1e-3, momentum=0.9, weight_decay=1e-4)
    f_main, f_L = FusedSGD([fused[i] for i in main_idx], **kw), FusedSGD([fused[i] for i in meh_idx], **kw)
    r_main, r_L = torch.optim.SGD([ref[i] for i in main_idx], **kw), torch.optim.SGD([ref[i] for i in meh_idx], **kw)
    s_main, s_L = torch.optim.SGD([ref32[i] for i in main_idx], **kw), torch.optim.SGD([ref32[i] for i in meh_idx], **kw)
    gen = torch.Generator().manual_seed(99)
    lr_main = 1e-3
    for step in range(4):
        grads = [torch.randn(p.shape, generator=gen) * (0.1 + step) for p in p0]
        if step == 1:
            f_main.device_lr(), f_L.device_lr()                                       # from now on the kernel reads the LR from HBM
        if step == 2:                                                                 # lr_config step: x0.1, main optimizer only
            lr_main = 1e-4
            for o in (f_main, r_main, s_main):
                o.param_groups[0]['lr'] = lr_main
        if step == 3:
            grads[3] = None                                                           # a parameter without gradient is skipped
        for i, gr in enumerate(grads):
            fused[i].grad = None if gr is None else gr.clone().cuda()
            ref[i].grad = None if gr is None else gr.clone().double()
            ref32[i].grad = None if gr is None else gr.clone()
        for o in (f_main, f_L, r_main, r_L, s_main, s_L):
            o.step()
        with torch.no_grad():
            omodel.sgd_step({str(i): orc[str(i)] for i in main_idx}, {str(i): grads[i] for i in main_idx}, bufs_main, lr=lr_main)
            omodel.sgd_step({str(i): orc[str(i)] for i in meh_idx}, {str(i): grads[i] for i in meh_idx}, bufs_L, lr=1e-3)
        torch.cuda.synchronize()
        for i in range(len(p0)):
            got = fused[i].detach().cpu()
            ideal = ref[i].detach()
            e_fused = float((got.double() - ideal).abs().max())
            e_torch = float((ref32[i].detach().double() - ideal).abs().max())
            scale = float(ideal.abs().max())
            assert e_fused <= max(2.0 * e_torch, 1e-7 * scale), (step, i, e_fused, e_torch)
            assert torch.allclose(got, ref32[i].detach(), rtol=1e-6, atol=1e-7 * scale), (step, i)
            assert torch.allclose(got, orc[str(i)], rtol=1e-6, atol=1e-7 * scale), (step, i)
        # momentum buffers too
        for o_f, o_s in ((f_main, s_main), (f_L, s_L)):
            for pf, ps in zip(o_f.param_groups[0]['params'], o_s.param_groups[0]['params']):
                if 'momentum_buffer' in o_s.state[ps]:
                    mb, ms = o_f.state[pf]['momentum_buffer'].cpu(), o_s.state[ps]['momentum_buffer']
                    assert torch.allclose(mb, ms, rtol=1e-6, atol=1e-7 * float(ms.abs().max())), step


def test_fused_sgd_eager_step_follows_lr_after_device_lr_was_enabled():
    """once device_lr() has been called the kernel prefers the device scalar; an EAGER step after the schedule changed
    group['lr'] (shape change between graph replays) must not use a stale value."""
    from aod_meh_hua_amd.optim import FusedSGD
    p = torch.nn.Parameter(torch.ones(1000, device='cuda'))
    opt = FusedSGD([p], lr=0.5, momentum=0.0, weight_decay=0.0)
    opt.device_lr()
    p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(p.detach(), torch.full_like(p, 0.5))
    opt.param_groups[0]['lr'] = 0.125                     # hook changes the LR; no device_lr() call follows (eager iteration)
    p.grad = torch.ones_like(p)
    opt.step()
    torch.cuda.synchronize()
    assert torch.allclose(p.detach(), torch.full_like(p, 0.375)), float(p[0])
