"""emu_twice.py with one batch per step: after step 1, how do the two runs' weights differ, and how do their step-2 gradients differ?"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
def fb(model, opt, opt_L, d):
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward()
    torch.cuda.synchronize()
    return float(out['loss'].detach()), float(lossL['loss'].detach())
runs = []
for k in range(2):
    model, opt, opt_L = mw.build()
    ps = opt.param_groups[0]['params'] + opt_L.param_groups[0]['params']
    nm = {id(p): n for n, p in model.named_parameters()}
    l1 = fb(model, opt, opt_L, mw.batch(0, 0)); opt.step(); opt_L.step(); torch.cuda.synchronize()
    w1 = [p.detach().clone() for p in ps]
    l2 = fb(model, opt, opt_L, mw.batch(1, 0))
    g2 = [p.grad.detach().clone() for p in ps]
    runs.append((l1, l2, w1, g2, [nm[id(p)] for p in ps]))
a, b = runs
print('losses step 1', a[0], b[0], ' step 2', a[1], b[1])
wd = sorted(((float((x - y).abs().max()), float((x - y).abs().max()) / (float(x.abs().max()) + 1e-20), n) for x, y, n in zip(a[2], b[2], a[4])), reverse=True)
print('weights after step 1, largest ABSOLUTE deviations (abs, rel, name):'); [print('  ', r) for r in wd[:6]]
nz = sum(1 for r in wd if r[0] > 0)
print('  parameters that differ at all:', nz, 'of', len(wd))
gd = sorted(((float((x - y).abs().max()) / (float(x.abs().max()) + 1e-20), n) for x, y, n in zip(a[3], b[3], a[4])), reverse=True)
print('step-2 gradients, largest relative deviations:'); [print('  ', r) for r in gd[:8]]
