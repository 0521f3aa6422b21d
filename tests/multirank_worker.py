"""Worker of tests/test_gpu_multirank.py (launched by torch.distributed.run with 2 ranks that share device 0 over gloo -- a 1-GPU box cannot
host two RCCL ranks; the code path is the product's: GradSync's flat in-place buffers, per-segment HIP graphs, eager all-reduces between).

Three data-parallel iterations, rank r training on its own batches, once through GraphedTrainStep (graph mode) and once eagerly
(the runner's order of operations); then on rank 0 a ONE-process emulation: gradients of both ranks' batches computed one after the
other, averaged, one optimizer step.  Checks: replicas bit-equal across ranks; DP == emulation up to fp32 summation order."""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as omodel      # noqa: E402  (seeded weights only)
from tests import synth                 # noqa: E402

STEPS = int(os.environ.get('MULTIRANK_STEPS', '3'))       # iterations of the graph / eager / emulation runs that are compared
LR = float(os.environ.get('MULTIRANK_LR', '2e-4'))


def build():
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    from aod_meh_hua_amd.optim import FusedSGD
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    model.load_state_dict(omodel.seeded_state_dict(cls_bias=-2.0), strict=True)
    model = model.cuda().train()
    head = model.bbox_head
    meh = set(id(p) for n in ('retina_L', 'L_convs') for p in getattr(head, n).parameters())
    main = [p for p in model.parameters() if p.requires_grad and id(p) not in meh]
    opt = FusedSGD(main, lr=LR, momentum=0.9, weight_decay=1e-4)
    opt_L = FusedSGD([p for p in model.parameters() if id(p) in meh], lr=LR, momentum=0.9, weight_decay=1e-4)
    return model, opt, opt_L


def batch(step, rank, B=2, H=128, W=None):
    W = W or H
    seed = 100 + 10 * step + rank
    gtb, gtl = synth.random_gts(B, H, W, seed=seed, gmin=1, gmax=3)
    return dict(img=synth.images(B, H, W, seed=seed).cuda(), img_metas=synth.metas(B, H, W), gt_bboxes=gtb, gt_labels=gtl)


def digest(model):
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def eager_dp_iter(model, opt, opt_L, gsync, data):
    """the runner's eager order under data parallelism (utils/Epoch_Based_Runner_Lambda.py run_iter): segmented backward, every segment's
    buckets all-reduced while the next segment computes"""
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.parallel import backward_and_sync
    main = opt.param_groups[0]['params']
    gsync.attach(main, segments=model.grad_segments(main))
    with AF.grad_cuts() as cuts:
        out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    assert len(cuts) == 3 and gsync.num_segments(main) == 4          # layer2 / layer3 / layer4 outputs; heads + neck, layer4, layer3, layer2
    opt.zero_grad()
    pending = backward_and_sync(gsync, main, out['loss'], cuts)
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad()
    lossL['loss'].backward()
    pending.wait()
    opt.step()
    gsync.all_reduce_grads(opt_L.param_groups[0]['params'])
    opt_L.step()


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    if os.environ.get('MULTIRANK_DETERMINISTIC', '0') == '1':       # ordered column sums (functional.set_deterministic)
        from aod_meh_hua_amd import functional as AF
        AF.set_deterministic(True)
    from aod_meh_hua_amd.graphs import GraphedTrainStep
    from aod_meh_hua_amd.parallel import GradSync, broadcast_model
    res = {}
    for mode in ('graph', 'eager'):
        model, opt, opt_L = build()
        broadcast_model(model)
        gsync = GradSync(bucket_mb=16)
        gs = GraphedTrainStep(model, opt, opt_L, grad_sync=gsync, warmup=1, Labeled=True, Pseudo=False) if mode == 'graph' else None
        for step in range(STEPS):
            d = batch(step, rank)
            if gs is not None:
                gs(d)
            else:
                eager_dp_iter(model, opt, opt_L, gsync, d)
        torch.cuda.synchronize()
        hs = [None] * world
        dist.all_gather_object(hs, digest(model))
        res[mode + '_replicas_equal'] = len(set(hs)) == 1
        res[mode + '_grad_is_flat_slice'] = all(p.grad is not None and p.grad.data_ptr() == p._aod_grad_view.data_ptr() for p in opt.param_groups[0]['params'])
        if rank == 0:
            res[mode] = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    # ranks that see DIFFERENT batch-shape sequences (keep-ratio VOC batches): each decides on its own whether it replays, captures or runs
    # eagerly (graphs.GraphedTrainStep.maybe); all three issue the same collectives, so the replicas must stay bit-equal and nothing may hang
    model, opt, opt_L = build()
    broadcast_model(model)
    gsync = GradSync(bucket_mb=16)
    gs = GraphedTrainStep(model, opt, opt_L, grad_sync=gsync, warmup=1, Labeled=True, Pseudo=False)
    shapes = {0: [(128, 128), (128, 128), (128, 160), (128, 128), (128, 128), (128, 160)],
              1: [(128, 128), (128, 160), (128, 160), (128, 160), (128, 128), (128, 160)]}[rank]
    modes = []
    for step, (H, W) in enumerate(shapes):
        d = batch(step, rank, H=H, W=W)
        if gs.maybe(d) is None:
            eager_dp_iter(model, opt, opt_L, gsync, d)
            modes.append('eager')
        else:
            modes.append('graph')
    torch.cuda.synchronize()
    hs = [None] * world
    dist.all_gather_object(hs, digest(model))
    res['mixed_replicas_equal'] = len(set(hs)) == 1
    ms = [None] * world
    dist.all_gather_object(ms, modes)
    res['mixed_modes'] = ms
    if rank == 0:
        # one process, both ranks' batches, mean gradient
        model, opt, opt_L = build()
        pm, pl = opt.param_groups[0]['params'], opt_L.param_groups[0]['params']
        for step in range(STEPS):
            gm, gl = [], []
            for r in range(world):
                d = batch(step, r)
                out, head_out, feat_out, prev = model.train_step(d, Labeled=True, Pseudo=False)
                opt.zero_grad()
                out['loss'].backward()
                gm.append([p.grad.detach().clone() for p in pm])
                lossL = model.train_step_L(prev, head_out, feat_out)
                opt_L.zero_grad()
                lossL['loss'].backward()
                gl.append([p.grad.detach().clone() for p in pl])
            for ps, gs_, o in ((pm, gm, opt), (pl, gl, opt_L)):
                for i, p in enumerate(ps):
                    p.grad = sum(g[i] for g in gs_) / world
                o.step()
        ref = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        for mode in ('graph', 'eager'):
            devs = {k: float((res[mode][k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)) for k in ref if ref[k].is_floating_point()}
            res[mode + '_vs_mean_gradient_run'] = max(devs.values())
            res[mode + '_worst_keys'] = sorted(devs, key=devs.get, reverse=True)[:4]
            del res[mode]
        moved = float((ref['bbox_head.retina_cls.weight'] - omodel.seeded_state_dict(cls_bias=-2.0)['bbox_head.retina_cls.weight']).abs().max())
        res['moved'] = moved
        print('MULTIRANK ' + json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
