"""every torch (aten) op that launches a device kernel in one eager train step, forward AND backward (autograd on the calling thread), with its
Python site and count -- the launch-bound glue left around the HIP kernels.  PREC=bf16|bf16x3"""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.utils._python_dispatch import TorchDispatchMode
from aod_meh_hua_amd import functional as AF
from aod_meh_hua_amd.models.dense_heads.L_anchor_head import PackedGT, pack_gts
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
data = dict(data, gt_bboxes=PackedGT(pack_gts(data['gt_bboxes'], data['gt_labels'], dev)), gt_labels=None)      # (what the graphed step feeds)
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.permute', 'aten.select', 'aten.slice', 'aten.as_strided', 'aten.alias',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten._local_scalar',
        'aten.empty', 'aten.lift_fresh', 'aten.is_', 'aten.narrow', 'aten.new_empty', 'aten._reshape_alias')
counts = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat = out if isinstance(out, (tuple, list)) else (out,)
            if any(torch.is_tensor(t) and t.is_cuda for t in flat):
                st = [f for f in traceback.extract_stack() if 'aod_meh_hua_amd' in f.filename]
                site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-2:]) if st else '(autograd)'
                counts[(name, site)] += 1
        return out
def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()
for _ in range(2): step()
with Log(): step()
print(sum(counts.values()), 'torch launches')
for (name, site), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{c:3d}  {name:34s} {site}')
