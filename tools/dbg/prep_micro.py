"""Time the batched parameter preparation (aod_param_prep) of the RetinaNet-R50 model: all layers stale -> one launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import functional as AF
model, cfg = B.build_model(torch.device('cuda'))
data = B.synth_batch(2, 256, 256, torch.device('cuda'), 0)
out, *_ = model.train_step(data, Labeled=True, Pseudo=False)        # registers every layer
torch.cuda.synchronize()
def once():
    for it in AF.PREP.order: it.ver = None
    AF.PREP.refresh()
for _ in range(3): once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): once()
e1.record(); torch.cuda.synchronize()
print(f'param_prep: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per refresh, {len(AF.PREP.order)} layers, {AF.PREP.nblocks} blocks')
