"""List the torch (non-library) device ops of one eager train+score step with their Python call sites (torch.profiler, with_stack)."""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
from aod_meh_hua_amd.scoring import score_batch  # noqa
def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=16):
    if not ka.key.startswith('aten::') or ka.device_time_total <= 0: continue
    if ka.key in ('aten::empty', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::clone', 'aten::reshape', 'aten::item'): continue
    site = next((s for s in ka.stack if 'aod_meh_hua_amd' in s or 'bench.py' in s), ka.stack[0] if ka.stack else '?')
    rows.append((ka.count, ka.key, ka.device_time_total, site.strip()[-120:]))
for n, name, us, site in sorted(rows, key=lambda r: -r[2])[:110]:
    print(f'{n:3d} {us:8.0f} us  {name:24s} {site}')
