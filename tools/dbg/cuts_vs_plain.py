"""one process: gradients of the plain backward vs the segmented backward (functional.grad_cuts) vs GradSync-attached flat slices, per parameter"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
model, opt, opt_L = mw.build()
pm = opt.param_groups[0]['params']
names = {id(p): n for n, p in model.named_parameters()}
d = mw.batch(0, 0)
def grads(mode):
    if mode == 'plain':
        out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
        opt.zero_grad(); out['loss'].backward()
    else:
        with AF.grad_cuts() as cuts:
            out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
        opt.zero_grad(); AF.backward_segments(out['loss'], cuts)
    torch.cuda.synchronize()
    return [p.grad.detach().clone() if p.grad is not None else None for p in pm], float(out['loss'])
g0, l0 = grads('plain')
g0b, _ = grads('plain')
g1, l1 = grads('cuts')
print('loss', l0, l1)
rows = []
for p, a, b, c in zip(pm, g0, g1, g0b):
    if a is None or b is None:
        rows.append((float('inf'), names[id(p)], 'missing', a is None, b is None)); continue
    e = float((a - b).abs().max() / (a.abs().max() + 1e-20)); e2 = float((a - c).abs().max() / (a.abs().max() + 1e-20))
    rows.append((e, names[id(p)], tuple(a.shape), e2))
rows.sort(key=lambda r: -r[0])
for r in rows[:15]: print(r)
