"""one process: the 3-iteration mean-gradient emulation of tests/multirank_worker.py run TWICE from the same seeded weights -- how far do two
runs of the same arithmetic drift apart (column sums go through fp32 atomics: arrival order)?  PREC=bf16|bf16x3"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
def run(steps=3, world=2):
    model, opt, opt_L = mw.build()
    pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
    hist = []
    for step in range(steps):
        gm, gl = [], []
        for r in range(world):
            d = mw.batch(step, r)
            out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
            opt.zero_grad(); out['loss'].backward()
            gm.append([p.grad.detach().clone() for p in pm])
            lossL = model.train_step_L(prev, head_out, feat_out)
            opt_L.zero_grad(); lossL['loss'].backward()
            gl.append([p.grad.detach().clone() for p in pl])
        for ps, gs_, o in ((pm, gm, opt), (pl, gl, opt_L)):
            for i, p in enumerate(ps):
                p.grad = sum(g[i] for g in gs_) / world
            o.step()
        torch.cuda.synchronize()
        hist.append({k: v.detach().float().clone() for k, v in model.state_dict().items()})
    return hist
W = int(os.environ.get('WORLD', '2'))
a, b = run(world=W), run(world=W)
for s in range(3):
    devs = {k: float((a[s][k] - b[s][k]).abs().max() / (a[s][k].abs().max() + 1e-12)) for k in a[s] if a[s][k].is_floating_point()}
    worst = sorted(devs, key=devs.get, reverse=True)[:3]
    print('after step', s + 1, 'worst relative deviation between two identical runs:', [(k, devs[k]) for k in worst])
