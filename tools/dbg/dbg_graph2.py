import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_graphs import _build, _batch
from aod_meh_hua_amd import functional as AF
b0 = _batch(31)
model, opt, opt_L = _build()
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
model.train_step(b0, Labeled=True, Pseudo=False)          # registers everything
w = model.bbox_head.retina_cls.weight
it = AF.PREP.items[id(w)]
def packed_err(ref): return float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - ref).abs().max())
# capture ONLY: refresh ; w += 1
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    AF.PREP.refresh(); w.data.add_(0.0)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    AF.PREP.refresh()
    w.data.add_(1.0)
with torch.no_grad(): w.copy_(sd0['bbox_head.retina_cls.weight'])
it.wf.zero_()
g.replay(); torch.cuda.synchronize()
print('after replay: wf vs w0', packed_err(sd0['bbox_head.retina_cls.weight']), ' wf vs w0+1', packed_err(sd0['bbox_head.retina_cls.weight'] + 1))
