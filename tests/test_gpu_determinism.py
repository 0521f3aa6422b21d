"""GPU: deterministic mode (aod_set_deterministic, functional.set_deterministic; the reference's `--deterministic`,
tools/train_RetinaNet.py:56-68).  Everything on the training path is order-deterministic by construction except the bias / BN-shift column
sums (fp32 atomics); in this mode they are ordered sums of per-workgroup partials (csrc/determinism.hip).  Two runs from the same seeded
weights must then end in the SAME bits -- eagerly and through the captured graphs, in both arithmetic modes -- where the default mode's
reference-precision runs drift apart within three iterations (tools/dbg/emu_twice.py)."""
import hashlib
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _digest(model):
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def _run(graph, steps=3):
    import multirank_worker as mw
    model, opt, opt_L = mw.build()
    gs = None
    if graph:
        from aod_meh_hua_amd.graphs import GraphedTrainStep
        gs = GraphedTrainStep(model, opt, opt_L, warmup=1, Labeled=True, Pseudo=False)
    for step in range(steps):
        d = mw.batch(step, step % 2)
        if gs is not None:
            gs(d)
            continue
        out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
        opt.zero_grad()
        out['loss'].backward()
        lossL = model.train_step_L(prev, head_out, feat_out)
        opt_L.zero_grad()
        lossL['loss'].backward()
        opt.step()
        opt_L.step()
    torch.cuda.synchronize()
    return _digest(model), {k: v.detach().float().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize('prec', ['bf16x3', 'bf16'])
def test_deterministic_mode_repeats_bit_for_bit(prec):
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd._C import lib
    AF.set_precision(prec)
    try:
        free, free_state = _run(False, steps=1)
        AF.set_deterministic(True)
        assert lib.aod_get_deterministic() == 1
        a, sa = _run(False)
        b, _ = _run(False)
        assert a == b, 'two eager runs in the deterministic mode ended in different bits'
        g1, sg = _run(True)
        g2, _ = _run(True)
        assert g1 == g2, 'two graph-replayed runs in the deterministic mode ended in different bits'
        # eager and replayed steps launch the same kernels: the same bits again
        assert g1 == a
        # ... and the mode changes the ORDER of the column sums, nothing else: one iteration equals the default mode's to fp32 rounding
        _, one = _run(False, steps=1)
        worst = max(float((one[k] - free_state[k]).abs().max() / (free_state[k].abs().max() + 1e-12)) for k in one if one[k].is_floating_point())
        assert worst < 2e-6, worst
    finally:
        AF.set_deterministic(False)
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
    assert lib.aod_get_deterministic() == 0


@pytest.mark.parametrize('prec', ['bf16x3', 'bf16'])
def test_l2norm_scale_gradient_is_ordered_in_the_deterministic_mode(prec):
    """SSD's L2Norm (config 0): its scale gradient is a column sum over every pixel -- per-element fp32 atomics by default, ordered per-block
    partials in the deterministic mode (csrc/ssd_ops.hip, csrc/x3_ops.hip)"""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.functional_ssd import l2norm
    AF.set_precision(prec)
    try:
        g = torch.Generator(device='cuda').manual_seed(6)
        B, C, H, W = 3, 512, 38, 37
        x = torch.randn(B, C, H, W, device='cuda', generator=g) * 3
        go = torch.randn(B, C, H, W, device='cuda', generator=g)
        rows = lambda t: (ho.x3_split(t.permute(0, 2, 3, 1).reshape(-1, C).contiguous()) if prec == 'bf16x3'
                          else t.permute(0, 2, 3, 1).reshape(-1, C).contiguous().bfloat16())
        w = (torch.rand(C, device='cuda', generator=g) * 10 + 15)

        def grad_w():
            xd = AF.as_nchw(rows(x), B, H, W).requires_grad_(True)
            wd = w.clone().requires_grad_(True)
            y = l2norm(xd, wd, 1e-10)
            y.backward(AF.as_nchw(rows(go), B, H, W))
            torch.cuda.synchronize()
            return wd.grad.clone(), xd.grad.clone()
        free_w, free_x = grad_w()
        AF.set_deterministic(True)
        a_w, a_x = grad_w()
        b_w, b_x = grad_w()
        assert torch.equal(a_w, b_w) and torch.equal(a_x, b_x)
        assert torch.equal(a_x, free_x)
        assert float((a_w - free_w).abs().max() / free_w.abs().max()) < 1e-5
    finally:
        AF.set_deterministic(False)
        AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))
