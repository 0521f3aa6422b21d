"""one process: gradients of batch B computed right after a pass on batch A (no optimizer step between) vs on a fresh model -- state that
leaks from one backward pass into the next shows up here"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
def grads(model, opt, opt_L, d):
    pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in pm + pl]
A, Bt = mw.batch(0, 0), mw.batch(0, 1)
m1 = mw.build(); names = [n for n, p in m1[0].named_parameters() if p.requires_grad]
gB_fresh = grads(*m1, Bt)
m2 = mw.build()
gA = grads(*m2, A)
gB_after = grads(*m2, Bt)
pm = m1[1].param_groups[0]['params'] + m1[2].param_groups[0]['params']
nm = {id(p): n for n, p in m1[0].named_parameters()}
rows = sorted(((float((a - b).abs().max()) / (float(a.abs().max()) + 1e-20), nm[id(p)]) for p, a, b in zip(pm, gB_fresh, gB_after)), reverse=True)
for r in rows[:10]: print(r)
