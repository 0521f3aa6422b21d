"""Which Python sites launch device-to-device copies (aten.copy_ / clone / _to_copy / cat) of >= 256 K elements in one eager train step +
scoring pass -- forward AND backward (autograd runs on the calling thread here, so the dispatch mode sees it).  PREC=bf16|bf16x3."""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from torch.utils._python_dispatch import TorchDispatchMode
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
counts = collections.Counter()
WATCH = ('aten.copy_', 'aten.clone', 'aten._to_copy', 'aten.cat', 'aten.contiguous', 'aten.add', 'aten.zeros', 'aten.zero_', 'aten.fill_')
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(WATCH):
            flat = out if isinstance(out, (tuple, list)) else (out,)
            big = [t for t in flat if torch.is_tensor(t) and t.is_cuda and t.numel() >= (1 << 18)]
            if big:
                st = [f for f in traceback.extract_stack() if 'aod_meh_hua_amd' in f.filename or f.filename.endswith('bench.py')]
                site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-3:]) if st else '?'
                counts[(name, site, big[0].numel() * big[0].element_size() >> 20)] += 1
        return out
def step():
    model.train()
    if os.environ.get('CUTS', '0') == '1':          # the segmented backward of graphs.GraphedTrainStep
        with AF.grad_cuts() as cuts:
            out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        opt.zero_grad(); AF.backward_segments(out['loss'], cuts)
    else:
        out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
        opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()
for _ in range(2): step()
with Log(): step()
for (name, site, mb), c in sorted(counts.items(), key=lambda kv: -kv[0][2] * kv[1]):
    print(f'{c:3d} x {mb:5d} MB  {name:28s} {site}')
