"""grad_ab.py on the multirank worker's model / batches (2 x 128 x 128): per-parameter gradient difference under an environment switch"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import multirank_worker as W
name = sys.argv[1]
model, opt, opt_L = W.build()
res = {}
for v in ('0', '1'):
    os.environ[name] = v
    d = W.batch(0, 0)
    model.zero_grad(set_to_none=True)
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    out['loss'].backward()
    torch.cuda.synchronize()
    res[v] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
rows = []
for n in res['0']:
    a, b = res['0'][n], res['1'][n]
    rows.append((float((a - b).abs().max() / (a.abs().max() + 1e-20)), n, float(a.abs().max())))
rows.sort(reverse=True)
for r in rows[:16]:
    print('%.3e  %-50s max|g| %.3e' % r)
