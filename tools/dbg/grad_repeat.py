"""one process: after NSTEP optimizer steps, the gradients of the same batch computed REP times with nothing changed in between -- run-to-run
noise of the backward pass itself, per parameter.  PREC=bf16|bf16x3 NSTEP=1"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
model, opt, opt_L = mw.build()
pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
nm = {id(p): n for n, p in model.named_parameters()}
def grads(d, step=False):
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward()
    if step:
        opt.step(); opt_L.step()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in pm + pl], float(out['loss'])
for s in range(int(os.environ.get('NSTEP', '1'))):
    grads(mw.batch(s, 0), step=True)
d = mw.batch(5, 0)
gs = [grads(d) for _ in range(3)]
print('losses', [g[1] for g in gs])
rows = []
for i, p in enumerate(pm + pl):
    a = gs[0][0][i]
    e = max(float((a - g[0][i]).abs().max()) for g in gs[1:]) / (float(a.abs().max()) + 1e-20)
    rows.append((e, nm[id(p)], tuple(a.shape)))
rows.sort(reverse=True)
for r in rows[:12]: print(r)
