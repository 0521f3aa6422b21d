import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
from tests import synth
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')
gold = np.load(os.path.join(G, 'losses.npz'))
li = synth.loss_inputs()
dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in li.items()}
noR, sums = ho.edl_focal_l1_fwd(dev['logits'], dev['labels'], dev['label_weights'], dev['bbox_pred'], dev['bbox_targets'], dev['bbox_weights'])
a, b = noR.cpu().numpy(), gold['loss_noR']
print('fwd per-row max rel err', float(np.max(np.abs(a - b) / (np.abs(b) + 1e-12))), 'tolerance 2e-5')
n = li['num_total_samples']
one = torch.full((1,), 1.0 / n, device='cuda')
gc, gb = ho.edl_focal_l1_bwd(dev['logits'], dev['labels'], dev['label_weights'], dev['bbox_pred'], dev['bbox_targets'], dev['bbox_weights'], one, one, None, 1.0 / 1024)
a, b = gc.cpu().numpy(), gold['grad_logits']
print('bwd max |err| / (5e-4 |ref| + 2e-7):', float(np.max(np.abs(a - b) / (5e-4 * np.abs(b) + 2e-7))), '(must be <= 1)')
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 16 * 36864
x = torch.randn(N, 20, device='cuda') * 2 - 3; lab = torch.randint(0, 21, (N,), device='cuda'); lw = torch.ones(N, device='cuda')
bp = torch.randn(N, 4, device='cuda'); bt = torch.randn(N, 4, device='cuda'); bw = torch.ones(N, 4, device='cuda')
print('fwd P3-level rows: %.1f us' % t(lambda: ho.edl_focal_l1_fwd(x, lab, lw, bp, bt, bw)))
print('bwd P3-level rows: %.1f us' % t(lambda: ho.edl_focal_l1_bwd(x, lab, lw, bp, bt, bw, one, one, None, 1.0 / 1024)))
