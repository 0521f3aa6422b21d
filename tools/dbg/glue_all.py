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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_input_shape=True, group_by_stack_n=24):
    if not ka.key.startswith('aten::') or ka.device_time_total <= 0:
        continue
    if ka.key in ('aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::clone', 'aten::reshape', 'aten::item', 'aten::zeros', 'aten::zeros_like',
                  'aten::stack', 'aten::zero_', 'aten::sub', 'aten::rsub'):
        continue                                    # wrappers: their leaf op (copy_, fill_, cat, ...) is listed
    st = [x for x in ka.stack if 'aod_meh_hua_amd' in x or 'bench.py' in x or 'glue_all' in x]
    site = ' <- '.join(x.strip().split('/')[-1][:60] for x in st[:3]) if st else (ka.stack[0].strip()[-100:] if ka.stack else '?')
    rows.append((ka.count, ka.key, ka.device_time_total, str(ka.input_shapes)[:60], site))
print(f'aten device time per step: {sum(r[2] for r in rows):.0f} us')
for n, name, us, shp, site in sorted(rows, key=lambda r: -r[2])[:120]:
    print(f'{n:3d} {us:8.1f} us  {name:16s} {shp:60s} {site}')
