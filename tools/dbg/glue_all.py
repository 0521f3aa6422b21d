"""Every NON-library device launch (torch aten kernels, memcpy / memset nodes) of one eager train + score step with its Python call site
(torch.profiler with_stack; backward ops carry the stack of the autograd Function's backward).  Answers: where do the copyBuffer / add /
fill launches of the replayed step come from?"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
pool = B.synth_batch(16, 512, 512, dev, 1)
import copy
pm = copy.deepcopy(model)
B.calibrate_head(pm, pool['img'])
ids = torch.arange(16, device=dev)


def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()
    pm.eval()
    with torch.no_grad():
        pm(img=[pool['img']], img_metas=[pool['img_metas']], image_ids=ids, **B.SCORE_KW)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.key.startswith('aten::'):
        continue
    if any(c.key.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        continue                                    # count leaf aten ops only
    st = [s for s in (ev.stack or []) if 'aod_meh_hua_amd' in s or 'bench.py' in s or 'glue_all' in s]
    site = st[0].strip()[-110:] if st else ((ev.stack or ['?'])[0].strip()[-110:])
    a = agg[(ev.key, site)]
    a[0] += 1; a[1] += ev.device_time_total
tot = sum(v[1] for v in agg.values())
print(f'aten device time per step: {tot:.0f} us in {sum(v[0] for v in agg.values())} ops')
for (k, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f'{n:3d} {us:8.1f} us  {k:26s} {site}')
