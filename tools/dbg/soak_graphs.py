"""Soak: 240 training iterations over six padded batch shapes that keep coming back (GraphedTrainStep.maybe: eager -> capture -> replay from the
cache) followed by 120 scoring batches through GraphedScore; prints device memory (allocated / reserved) and host RSS every 40 iterations --
a leak in the graph caches, the scratch dictionaries or the ctypes staging shows as a slope.   run on GPU:  python tools/dbg/soak_graphs.py"""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch      # noqa: E402
from tests import synth      # noqa: E402
from tests.test_gpu_graphs import _build, _eager_iter      # noqa: E402
from aod_meh_hua_amd.graphs import GraphedScore, GraphedTrainStep      # noqa: E402

shapes = [(160, 272), (176, 288), (200, 336), (192, 192), (224, 320), (160, 240)]


def batch(seed, hw):
    H, W = hw
    gtb, gtl = synth.random_gts(2, H, W, seed=seed, gmin=1, gmax=3)
    return dict(img=synth.images(2, H, W, seed=seed).cuda(), img_metas=synth.metas(2, H, W), gt_bboxes=gtb, gt_labels=gtl)


model, opt, opt_L = _build(lr=1e-5)
gs = GraphedTrainStep(model, opt, opt_L, warmup=1, Labeled=True, Pseudo=False)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
how = {'eager': 0, 'graph': 0}
for it in range(240):
    d = batch(1000 + it, shapes[(it // 2) % len(shapes)])
    o = gs.maybe(d)
    if o is None:
        how['eager'] += 1
        loss = _eager_iter(model, opt, opt_L, d)[0]
    else:
        how['graph'] += 1
        loss = float(o['loss'])
    if it % 40 == 39:
        torch.cuda.synchronize()
        print(f'train {it + 1:4d}: loss {loss:10.4f}  allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  '
              f'graphs {len(gs.cache)}  host RSS {rss():8.1f} MiB  {how}', flush=True)
assert loss == loss, 'loss is NaN'
model.eval()
kw = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False,
          showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
gsc = GraphedScore(model, **kw)
for it in range(120):
    d = batch(5000 + it, shapes[it % len(shapes)])
    ids = torch.tensor([2 * it, 2 * it + 1], device='cuda')
    _, unc = gsc(d['img'], d['img_metas'], ids, defer=(it % 3 == 0))
    if it % 40 == 39:
        gsc.sync(); torch.cuda.synchronize()
        print(f'score {it + 1:4d}: allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  graphs {len(gsc.cache)}  '
              f'host RSS {rss():8.1f} MiB', flush=True)
print('done')
