"""two models trained in lockstep from the same seeded weights (they differ only through the arrival order of fp32 atomics): at the step where
their parameters jump apart, which forward outputs / losses / gradients differ first?  PREC=bf16x3 STEPS=3"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
ms = [mw.build() for _ in range(2)]
nm = {id(p): n for n, p in ms[0][0].named_parameters()}
def rel(a, b): return float((a.float() - b.float()).abs().max()) / (float(a.float().abs().max()) + 1e-20)
STEPS = int(os.environ.get('STEPS', '3'))
for step in range(STEPS):
    d = mw.batch(step, 0)
    res = []
    for model, opt, opt_L in ms:
        out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
        info, cls, box = head_out[0], head_out[1], head_out[2]
        for t in list(cls) + list(box): t.retain_grad()
        opt.zero_grad(); out['loss'].backward()
        lossL = model.train_step_L(prev, head_out, feat_out)
        opt_L.zero_grad(); lossL['loss'].backward()
        torch.cuda.synchronize()
        ps = opt.param_groups[0]['params'] + opt_L.param_groups[0]['params']
        res.append(dict(cls=[t.detach().clone() for t in cls], box=[t.detach().clone() for t in box], gcls=[t.grad.clone() for t in cls], gbox=[t.grad.clone() for t in box],
                        lv={k: float(v) for k, v in out['log_vars'].items()}, g=[p.grad.detach().clone() for p in ps], names=[nm.get(id(p), '?') for p in ps],
                        labels=[t.clone() for t in head_out[4]], bt=[t.clone() for t in head_out[6]]))
        opt.step(); opt_L.step()
    a, b = res
    print(f'--- step {step + 1}: log_vars', {k: (a['lv'][k], b['lv'][k]) for k in a['lv'] if a['lv'][k] != b['lv'][k]} or 'identical')
    print('  cls_scores dev', [f'{rel(x, y):.1e}' for x, y in zip(a['cls'], b['cls'])], ' bbox_preds dev', [f'{rel(x, y):.1e}' for x, y in zip(a['box'], b['box'])])
    print('  d loss / d cls dev', [f'{rel(x, y):.1e}' for x, y in zip(a['gcls'], b['gcls'])], ' d loss / d bbox dev', [f'{rel(x, y):.1e}' for x, y in zip(a['gbox'], b['gbox'])])
    print('  targets equal', all(torch.equal(x, y) for x, y in zip(a['labels'], b['labels'])), all(torch.equal(x, y) for x, y in zip(a['bt'], b['bt'])))
    # where bbox gradients differ: the prediction - target values there
    for l, (x, y) in enumerate(zip(a['gbox'], b['gbox'])):
        dmask = (x != y)
        if dmask.any():
            n = int(dmask.sum())
            big = (x - y).abs() > 1e-3 * float(x.abs().max())
            print(f'    level {l}: {n} bbox-gradient elements differ, {int(big.sum())} of them by > 1e-3 of the largest gradient')
    gd = sorted(((rel(x, y), n) for x, y, n in zip(a['g'], b['g'], a['names'])), reverse=True)
    print('  parameter gradients:', [(n, f'{e:.1e}') for e, n in gd[:5]])
