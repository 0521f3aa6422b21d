"""2 ranks on device 0 over gloo: ONE eager data-parallel iteration; rank 0 compares its all-reduced gradients and updated parameters with a
one-process mean-gradient emulation, per parameter.  torchrun --nproc-per-node 2 tools/dbg/dp_debug_worker.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch.distributed as dist
from aod_meh_hua_amd import functional as AF
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
import multirank_worker as mw
from aod_meh_hua_amd.parallel import GradSync, broadcast_model
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
model, opt, opt_L = mw.build()
broadcast_model(model)
gsync = GradSync(bucket_mb=16)
pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
nm = {id(p): n for n, p in model.named_parameters()}
# --- DP iteration, but stop before the optimizer steps to look at the gradients
from aod_meh_hua_amd.parallel import backward_and_sync
d = mw.batch(0, rank)
gsync.attach(pm, segments=model.grad_segments(pm))
with AF.grad_cuts() as cuts:
    out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
opt.zero_grad()
pending = backward_and_sync(gsync, pm, out['loss'], cuts)
lossL = model.train_step_L(prev, head_out, feat_out)
opt_L.zero_grad(); lossL['loss'].backward()
pending.wait()
gsync.all_reduce_grads(pl)
torch.cuda.synchronize()
g_dp = [p.grad.detach().clone() for p in pm + pl]
opt.step(); opt_L.step(); torch.cuda.synchronize()
w_dp = [p.detach().clone() for p in pm + pl]
if rank == 0:
    m2, o2, oL2 = mw.build()
    q = o2.param_groups[0]['params'] + oL2.param_groups[0]['params']
    gs = []
    for r in range(world):
        dd = mw.batch(0, r)
        out, head_out, feat_out, prev = m2.train_step(dd, Labeled=True, Pseudo=False)
        o2.zero_grad(); out['loss'].backward()
        lossL = m2.train_step_L(prev, head_out, feat_out)
        oL2.zero_grad(); lossL['loss'].backward()
        gs.append([p.grad.detach().clone() for p in q])
    g_em = [sum(g[i] for g in gs) / world for i in range(len(q))]
    for i, p in enumerate(q): p.grad = g_em[i]
    o2.step(); oL2.step(); torch.cuda.synchronize()
    rows = []
    for i, p in enumerate(pm + pl):
        ge = float((g_dp[i] - g_em[i]).abs().max()) / (float(g_em[i].abs().max()) + 1e-20)
        we = float((w_dp[i] - q[i].detach()).abs().max()) / (float(q[i].detach().abs().max()) + 1e-20)
        g0e = float((gs[0][i] * 0 + g_dp[i] - gs[0][i]).abs().max()) / (float(gs[0][i].abs().max()) + 1e-20)
        rows.append((ge, we, g0e, nm[id(p)], float(g_em[i].abs().max()), float(q[i].detach().abs().max())))
    rows.sort(reverse=True)
    print('(grad dev, weight dev, dp-grad vs rank-0-only grad, name, |g|max, |w|max)')
    for r in rows[:14]: print(r)
dist.barrier(); dist.destroy_process_group()
