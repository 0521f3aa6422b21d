"""copy_sites.py for the CAPTURED step: the torch ops on >= 4 M-element tensors that run while graphs.GraphedTrainStep warms up / captures"""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.utils._python_dispatch import TorchDispatchMode
from aod_meh_hua_amd import functional as AF
from aod_meh_hua_amd.graphs import GraphedTrainStep
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
counts = collections.Counter()
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.permute', 'aten.select', 'aten.slice', 'aten.as_strided', 'aten.alias',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten._local_scalar',
        'aten.empty', 'aten.lift_fresh', 'aten.is_', 'aten.narrow')
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat = list(out) if isinstance(out, (tuple, list)) else [out]
            flat += [a for a in args if torch.is_tensor(a)]
            big = [t for t in flat if torch.is_tensor(t) and t.is_cuda and t.numel() >= (1 << 22)]
            if big:
                st = [f for f in traceback.extract_stack() if 'aod_meh_hua_amd' in f.filename or f.filename.endswith('bench.py')]
                site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-4:]) if st else '?'
                counts[(name, site, big[0].numel() * big[0].element_size() >> 20)] += 1
        return out
g = GraphedTrainStep(model, opt, opt_L, Labeled=True, Pseudo=False)
with Log():
    for _ in range(4): g(data)
torch.cuda.synchronize()
for (name, site, mb), c in sorted(counts.items(), key=lambda kv: -kv[0][2] * kv[1])[:40]:
    print(f'{c:3d} x {mb:5d} MB  {name:28s} {site}')
