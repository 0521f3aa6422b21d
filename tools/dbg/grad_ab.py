"""All parameter gradients of one eager training iteration under two settings of an environment switch (argv: NAME): max relative difference
per parameter, largest first."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
name = sys.argv[1]
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
data = B.synth_batch(4, 256, 256, dev, 0)
res = {}
for v in ('0', '1'):
    os.environ[name] = v
    model.train()
    model.zero_grad(set_to_none=True)
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    out['loss'].backward()
    torch.cuda.synchronize()
    res[v] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
rows = []
for n in res['0']:
    a, b = res['0'][n], res['1'][n]
    rows.append((float((a - b).abs().max() / (a.abs().max() + 1e-20)), n, float(a.abs().max())))
rows.sort(reverse=True)
for r in rows[:25]:
    print('%.3e  %-50s max|g| %.3e' % r)
