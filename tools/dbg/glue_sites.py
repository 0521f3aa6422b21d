"""Call sites of the torch (aten) device ops in the FORWARD half of one eager train step (TorchDispatchMode: thread-local, so the autograd
thread's ops are not listed -- they mirror these)."""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.permute', 'aten.select', 'aten.slice', 'aten.as_strided', 'aten.alias',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten._local_scalar',
        'aten.empty', 'aten.lift_fresh', 'aten.is_')
counts = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat = out if isinstance(out, (tuple, list)) else (out,)
            if any(torch.is_tensor(t) and t.is_cuda for t in flat):
                st = [f for f in traceback.extract_stack() if 'aod_meh_hua_amd' in f.filename or f.filename.endswith('bench.py')]
                site = f'{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].line[:70]}' if st else '?'
                counts[(name, site)] += 1
        return out
def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16'))
import copy
pm = copy.deepcopy(model); B.calibrate_head(pm, data['img'])
def score():
    pm.eval()
    with torch.no_grad():
        pm(img=[data['img']], img_metas=[data['img_metas']], image_ids=torch.arange(16, device=dev), **B.SCORE_KW)
for _ in range(2): step(); score()
torch.cuda.synchronize()
with Log():
    step()
torch.cuda.synchronize()
print('---- train step'); 
for (name, site), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{n:3d} {name:32s} {site}')
counts.clear()
with Log():
    score()
torch.cuda.synchronize()
print('---- scoring pass')
for (name, site), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{n:3d} {name:32s} {site}')
