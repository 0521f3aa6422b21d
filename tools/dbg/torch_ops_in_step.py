"""Every torch op that launches something while graphs.GraphedTrainStep / graphs.GraphedScore run their eager + capture passes (the 'crumbs' of
profiles/r06_step_trace.txt: copies, fills, small elementwise / reduce kernels between the library's launches), by call site.
  gpurun -- 'python tools/dbg/torch_ops_in_step.py > gpurun_out/torch_ops_in_step.txt'"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from aod_meh_hua_amd import functional as AF  # noqa: E402
from aod_meh_hua_amd.graphs import GraphedScore, GraphedTrainStep  # noqa: E402

AF.set_precision(os.environ.get('PREC', 'bf16x3'))
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 20)
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.permute', 'aten.select', 'aten.slice', 'aten.as_strided', 'aten.alias',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape', 'aten.transpose', 'aten.unbind', 'aten.split', 'aten._local_scalar',
        'aten.empty', 'aten.lift_fresh', 'aten.is_', 'aten.narrow', 'aten.chunk', 'aten.unflatten', 'aten.flatten', 'aten.movedim', 'aten.new_empty',
        'aten.sym_', 'aten.stride', 'aten.size', 'aten.numel', 'aten.dim', 'aten.is_contiguous', 'aten._to_copy.default_meta')


ONLY_CAPTURED = os.environ.get('ONLY_CAPTURED', '1') != '0'      # 1: only the ops that end up INSIDE a graph


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat = list(out) if isinstance(out, (tuple, list)) else [out]
            flat += [a for a in args if torch.is_tensor(a)]
            cu = [t for t in flat if torch.is_tensor(t) and t.is_cuda]
            if cu and (not ONLY_CAPTURED or torch.cuda.is_current_stream_capturing()):
                st = [f for f in traceback.extract_stack() if 'aod_meh_hua_amd' in f.filename or f.filename.endswith('bench.py')]
                site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-4:]) if st else '?'
                n = max(t.numel() * t.element_size() for t in cu)
                self.counts[(name, site, n)] += 1
        return out


def report(title, log, calls):
    print(f'== {title}: ops per call (logged over {calls} calls)')
    tot = 0
    for (name, site, n), c in sorted(log.counts.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f'{c / calls:5.1f} x {n:>11d} B  {name:34s} {site}')
        tot += c
    print(f'   total {tot / calls:.1f} per call')


import copy  # noqa: E402

pool = B.synth_batch(16, 512, 512, dev, 1020)
pool_model = copy.deepcopy(model)
B.calibrate_head(pool_model, pool['img'])
g = GraphedTrainStep(model, opt, opt_L, Labeled=True, Pseudo=False)
lg = Log()
with lg:
    for _ in range(3):
        g(data)                      # eager warm-up, capture, replays as the class schedules them
torch.cuda.synchronize()
report('GraphedTrainStep, first three calls (eager warm-up, capture, replay: ops outside the graph repeat on every call)', lg, 3)
lr = Log()
with lr:
    for _ in range(4):
        g(data)
torch.cuda.synchronize()
report('GraphedTrainStep, steady state (what still runs from Python around the replays)', lr, 4)
gs = GraphedScore(pool_model, **{k: v for k, v in B.SCORE_KW.items() if k != 'return_loss'})
ids = torch.arange(16, device=dev)
ls = Log()
with ls:
    for _ in range(3):
        gs(pool['img'], pool['img_metas'], ids)
torch.cuda.synchronize()
report('GraphedScore, first three calls', ls, 3)
lt = Log()
with lt:
    for _ in range(4):
        gs(pool['img'], pool['img_metas'], ids)
torch.cuda.synchronize()
report('GraphedScore, steady state', lt, 4)
