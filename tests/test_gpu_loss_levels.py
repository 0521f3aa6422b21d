"""The level-fused loss launches (aod_edl_focal_l1_levels_fwd / _bwd, aod_meh_loss_levels_fwd / _bwd: every pyramid level of
loss_single / loss_single_L -- /root/reference's mmdet/models/dense_heads/Lambda_L2.py:105-121,235-241 through multi_apply at
L_anchor_head.py:306-314,322-327 -- in one launch per pass) against the per-level launches they replace: identical bits, at the kernel
boundary (ragged and empty levels, 20 / 80 / 81 classes) and through the whole train iteration."""
import os

import pytest
import torch

from oracle import model as omodel
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('C,A,pixels', [(20, 9, (4096, 1000, 256, 0, 7)), (80, 9, (1500, 37, 1)), (81, 4, (64, 333)), (20, 1, (5,))])
def test_level_fused_loss_launches_equal_the_per_level_launches(C, A, pixels):
    from aod_meh_hua_amd import hipops as ho
    g = torch.Generator(device='cuda').manual_seed(5 + C + len(pixels))
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    rows = [p * A for p in pixels]
    R, L = sum(rows), len(rows)
    cls = rnd(R, C) * 2.0
    labels = torch.randint(0, C + 1, (R,), device='cuda', generator=g)
    lw = (torch.rand(R, device='cuda', generator=g) > 0.1).float()
    bp, bt = rnd(R, 4), rnd(R, 4)
    bw = (labels < C).float()[:, None].expand(R, 4).contiguous()
    noR, sums = ho.edl_focal_l1_levels_fwd(cls, labels, lw, bp, bt, bw, rows)
    g_sums = torch.rand(3, L, device='cuda', generator=g) + 0.5
    for g_rows in (None, rnd(R)):
        gc, gb = torch.full((R // A, A * C), 7.0, device='cuda'), torch.full((R // A, A * 4), 7.0, device='cuda')
        ho.edl_focal_l1_levels_bwd(cls, labels, lw, bp, bt, bw, rows, g_sums, g_rows, gc, gb, A)
        r = 0
        for l, n in enumerate(rows):
            sl = slice(r, r + n)
            noR_l, sums_l = ho.edl_focal_l1_fwd(cls[sl], labels[sl], lw[sl], bp[sl], bt[sl], bw[sl])
            assert torch.equal(noR[sl], noR_l) and torch.equal(sums[:, l], sums_l), (l, sums[:, l], sums_l)
            if n:
                gn = g_sums[2, l:l + 1].contiguous() if g_rows is None else g_rows[sl].contiguous()
                gc_l, gb_l = ho.edl_focal_l1_bwd(cls[sl], labels[sl], lw[sl], bp[sl], bt[sl], bw[sl], g_sums[0, l:l + 1].contiguous(),
                                                 g_sums[1, l:l + 1].contiguous(), gn, 0.0, g_noR_is_scalar=g_rows is None, A=A)
                assert torch.equal(gc[r // A:(r + n) // A], gc_l) and torch.equal(gb[r // A:(r + n) // A], gb_l), l
            r += n
    # the divided form: Q = sums / (num_total_samples | level rows) inside the reduction, g / divisor inside the backward kernel -- the IEEE
    # quotients of the tensor divisions they replace (L_anchor_head.py:266-288,300-303; SSL_Lambda.py:136-141)
    num_pos = torch.tensor([3, 0, 17, 1], dtype=torch.int32, device='cuda')
    if all(rows):
        noR2, Q, D, nt = ho.edl_focal_l1_levels_fwd(cls, labels, lw, bp, bt, bw, rows, num_pos=num_pos)
        assert float(nt) == 22.0 and torch.equal(noR2, noR)
        Dref = torch.tensor([[22.0] * L, [22.0] * L, [float(n) for n in rows]], device='cuda')
        assert torch.equal(D, Dref) and torch.equal(Q, sums / Dref)
        gc1, gb1 = torch.empty(R // A, A * C, device='cuda'), torch.empty(R // A, A * 4, device='cuda')
        gc2, gb2 = torch.empty_like(gc1), torch.empty_like(gb1)
        ho.edl_focal_l1_levels_bwd(cls, labels, lw, bp, bt, bw, rows, g_sums, None, gc1, gb1, A, divisors=D)
        ho.edl_focal_l1_levels_bwd(cls, labels, lw, bp, bt, bw, rows, (g_sums / D).contiguous(), None, gc2, gb2, A)
        assert torch.equal(gc1, gc2) and torch.equal(gb1, gb2)
    # MEH loss: lam per anchor row, weights = column 0 of bbox_w
    lam = rnd(R).abs()
    out = ho.meh_loss_levels_fwd(lam, noR, bw, rows)
    gm = torch.rand(L, device='cuda', generator=g) + 0.5
    gl = torch.full((R // A, A), 7.0, device='cuda')
    ho.meh_loss_levels_bwd(lam, noR, bw, rows, gm, gl, A)
    r = 0
    for l, n in enumerate(rows):
        sl = slice(r, r + n)
        if n:
            assert torch.equal(out[l:l + 1], ho.meh_loss_fwd(lam[sl], noR[sl], bw[sl])), l
            assert torch.equal(gl[r // A:(r + n) // A], ho.meh_loss_bwd(lam[sl], noR[sl], bw[sl], gm[l:l + 1].contiguous(), A=A)), l
        else:
            assert float(out[l]) == 0.0
        r += n


def test_train_iteration_with_level_fused_losses_equals_the_per_level_form(monkeypatch):
    """run_iter's two passes (Epoch_Based_Runner_Lambda.py:20-38): losses, loss rows and every gradient, both forms of the loss launches,
    in the deterministic column-sum mode (bias gradients are fp32 atomics otherwise)."""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(), strict=True)
    model = model.cuda().train()
    B, H, W = 3, 160, 224
    img = synth.images(B, H, W, seed=11)
    gtb, gtl = synth.random_gts(B, H, W, seed=12, gmin=1, gmax=4)
    data = dict(img=img.cuda(), img_metas=synth.metas(B, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    names = [n for n, _ in model.named_parameters()]
    calls = []
    orig = ho.call
    monkeypatch.setattr(ho, 'call', lambda name, *a: (calls.append(name), orig(name, *a))[1])
    res = {}
    ho.set_deterministic(True)
    try:
        for mode in (True, False):
            monkeypatch.setattr(AF, 'LOSS_LEVELS', mode)
            calls.clear()
            model.zero_grad(set_to_none=True)
            out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
            out['loss'].backward()
            g_main = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            model.zero_grad(set_to_none=True)
            outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
            outL['loss'].backward()
            g_L = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            n_level = sum(c.startswith(('aod_edl_focal_l1_levels', 'aod_meh_loss_levels')) for c in calls)
            n_single = sum(c in ('aod_edl_focal_l1_fwd', 'aod_edl_focal_l1_bwd', 'aod_meh_loss_fwd', 'aod_meh_loss_bwd') for c in calls)
            assert (n_level, n_single) == ((4, 0) if mode else (0, 20)), (mode, n_level, n_single)
            res[mode] = (out['loss'].detach().clone(), {k: v.detach().clone() for k, v in out['log_vars'].items()} if isinstance(out.get('log_vars'), dict) else {},
                         [t.detach().clone() for t in prev], outL['loss'].detach().clone(), g_main, g_L)
    finally:
        ho.set_deterministic(False)
    a, b = res[True], res[False]
    # (the total is sum(row sums of the [3, L] term matrix) in the fused form, (cls + bbox) + noR of per-name sums otherwise: last-bit differences)
    assert abs(float(a[0]) - float(b[0])) <= 2e-7 * abs(float(b[0])) and torch.equal(a[3], b[3]), (a[0], b[0], a[3], b[3])
    assert a[1].keys() == b[1].keys() and all(abs(float(a[1][k]) - float(b[1][k])) <= 2e-7 * abs(float(b[1][k])) for k in a[1]), (a[1], b[1])
    assert len(a[2]) == len(b[2]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    for ga, gb_ in ((a[4], b[4]), (a[5], b[5])):
        assert ga.keys() == gb_.keys() and len(ga) >= 10
        bad = [n for n in names if n in ga and not torch.equal(ga[n], gb_[n])]
        assert not bad, bad[:8]
