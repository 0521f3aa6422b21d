"""one process (gloo, world size 1): gradients through the data-parallel path (GradSync flat slices, segmented backward) vs the plain backward"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29561')
import torch.distributed as dist
dist.init_process_group('gloo', rank=0, world_size=1)
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
from aod_meh_hua_amd.parallel import GradSync, backward_and_sync
model, opt, opt_L = mw.build()
pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
names = {id(p): n for n, p in model.named_parameters()}
d = mw.batch(0, 0)
def plain():
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in pm + pl]
g0 = plain()
g0b = plain()
gsync = GradSync(bucket_mb=16)
def dp():
    gsync.attach(pm, segments=model.grad_segments(pm))
    with AF.grad_cuts() as cuts:
        out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
    opt.zero_grad()
    pending = backward_and_sync(gsync, pm, out['loss'], cuts)
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward()
    pending.wait()
    gsync.all_reduce_grads(pl)
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in pm + pl]
g1 = dp()
g2 = dp()
rows = []
for p, a, b, c, e in zip(pm + pl, g0, g1, g0b, g2):
    s = float(a.abs().max()) + 1e-20
    rows.append((float((a - b).abs().max()) / s, names[id(p)], tuple(a.shape), float((a - c).abs().max()) / s, float((a - e).abs().max()) / s))
rows.sort(key=lambda r: -r[0])
print('(dp vs plain, name, shape, plain vs plain again, second dp vs plain)')
for r in rows[:12]: print(r)
dist.destroy_process_group()
