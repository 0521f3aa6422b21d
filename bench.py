#!/usr/bin/env python
"""Benchmark of the MEH/HUA hot path on MI355X (contract: prompt section 'bench.py').

Workload (BASELINE.json configs[1]): RetinaNet-R50-FPN + MEH/HUA, synthetic VOC 512x512,
20 classes, 16 images per GPU.  One "step" = one AL work unit over one batch of synthetic images resident
in HBM: the full training iteration of MyEpochBasedRunnerLambda.run_iter (main forward + backward + SGD,
then MEH forward + backward + SGD) PLUS the HUA scoring pass over a batch of the same size
(forward + MEH forward + top-k + NMS + Dirichlet sampling + aggregation -> one score per image), i.e. the
metric "images/sec train + HUA-score".  value = images through both phases / time.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is measured live (HIP events on the launch stream around every
conv launch of one extra instrumented step); `cpu_baseline` times the CPU oracle (a faithful port of the
reference's CPU path, oracle/model.py) on a bounded sample on rank 0 at N == 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--mode', default='train+score', choices=['train', 'score', 'train+score'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--shapes', default=None, help='write the per-shape conv breakdown of the instrumented step to this file')
    ap.add_argument('--no-graph', action='store_true', help='enqueue every kernel from Python instead of replaying HIP graphs')
    return ap.parse_args()


def synth_batch(B, H, W, device, seed):
    """img ~ N(0,1); G ~ U{1..5}; w,h ~ U(32,384) clipped inside; labels ~ U{0..19}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, H, W, generator=g)
    boxes, labels = [], []
    for _ in range(B):
        G = int(torch.randint(1, 6, (1,), generator=g))
        wh = torch.rand(G, 2, generator=g) * (384 - 32) * (H / 512.0) + 32 * (H / 512.0)
        xy = torch.rand(G, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh).clamp(min=0)
        boxes.append(torch.cat([xy, (xy + wh).clamp(max=float(H))], 1))      # ground truth stays on the host like a data loader's
        labels.append(torch.randint(0, 20, (G,), generator=g))
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=np.ones(4, np.float32), flip=False)
             for _ in range(B)]
    return dict(img=img.to(device), img_metas=metas, gt_bboxes=boxes, gt_labels=labels)


def build_model(device, seed=20):
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')        # no network: random-init weights of the named architecture
    torch.manual_seed(seed)
    model = build_detector(cfg.model)
    model.init_weights()
    with torch.no_grad():
        model.bbox_head.retina_L.bias.fill_(0.1)
    return model.to(device), cfg


def calibrate_head(model, img, target_frac=0.005, fg_thr=0.3):
    """"Trained-like" classification head: random-init weights give max-softmax ~ 0.05 everywhere, so every image would score 0
    and the HUA sampler would idle.  Scale retina_cls (weight and bias: the logits scale exactly) by the factor k at which `target_frac` of
    all anchors of this batch exceed the foreground threshold 0.3 (Lambda_L2.py:349,497-502).  Set-up only (torch ops, not timed)."""
    from aod_meh_hua_amd.scoring import nhwc_view
    head = model.bbox_head
    model.eval()
    with torch.no_grad():
        cls, _ = head.forward(model.extract_feat(img))
        x = torch.cat([nhwc_view(c.float(), head.cls_out_channels) for c in cls], 1)
        frac = lambda k: float((torch.softmax(x * k, -1).amax(-1) > fg_thr).float().mean())
        lo, hi = 1.0, 2.0
        while frac(hi) < target_frac and hi < 1e6:
            lo, hi = hi, hi * 2
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if frac(mid) < target_frac else (lo, mid)
        head.retina_cls.weight.mul_(hi)
        head.retina_cls.bias.mul_(hi)
    return hi, frac(hi)


def hua_stats(model, pool, score_kw, dev, reps=20):
    """Pairs / objects per image of the scoring batch and the duration of aod_hua_score (HIP events on the launch stream; its three
    launches pairs -> sample -> reduce) -- SURVEY 8d asks for P-bar, O-bar and variates/s beside the number."""
    from aod_meh_hua_amd import scoring
    head = model.bbox_head
    model.eval()
    with torch.no_grad():
        feats = model.extract_feat(pool['img'])
        outs = head.forward(feats)
        Ls = head.forward_L(feats)
        kw = {k: v for k, v in score_kw.items() if k not in ('return_loss', 'rescale')}
        _, unc, it = head.get_bboxes(*outs, pool['img_metas'], rescale=True, with_nms=True, L_scores=Ls, _return_internals=True, **kw)
        B = unc.shape[0]
        ids = torch.arange(B, device=dev, dtype=torch.int64)
        args = (it['cand'], it['dets'], it['num'], ids, head.test_cfg.max_per_img)
        _, pc, _ = scoring.hua_score(*args, want_pairs=True)
        nobj = (it['dets'][..., 4] > 0.3).sum(1)
        scoring.hua_score(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            scoring.hua_score(*args)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    pairs = int(pc.sum())
    nd = head.cls_out_channels
    return dict(pairs_per_img=round(pairs / B, 1), objects_per_img=round(float(nobj.float().mean()), 1),
                nonzero_scores=int((unc > 0).sum()), hua_us_per_batch=round(us, 1), us_per_pair=round(us / max(pairs, 1), 4),
                gamma_variates_per_s=round(pairs * 500 * nd / (us * 1e-6), 0) if pairs else 0.0)


def make_optimizers(model, cfg):
    """apis/train_Lambda.py:54-61: main SGD without the MEH parameters + optimizer_L over the MEH parameters."""
    from aod_meh_hua_amd.optim import FusedSGD
    o = cfg.optimizer
    meh = [p for n, p in model.named_parameters() if ('retina_L' in n or 'L_convs' in n)]
    ids = {id(p) for p in meh}
    main = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    return (FusedSGD(main, lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay),
            FusedSGD(meh, lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay))


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # debugging aid only: AOD_BENCH_ONE_GPU=1 runs every rank on device 0 over gloo (functional check of the multi-rank path on a
    # single-GPU box); the driver's multi-GPU runs use one GPU per rank over RCCL
    one_gpu = os.environ.get('AOD_BENCH_ONE_GPU') == '1'
    local = 0 if one_gpu else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.parallel import GradSync, broadcast_model, gather_scores
    model, cfg = build_model(dev)
    broadcast_model(model)
    opt, opt_L = make_optimizers(model, cfg)
    gsync = GradSync()
    B, H = args.batch, args.size
    data = synth_batch(B, H, H, dev, seed=20 + rank)
    pool = synth_batch(B, H, H, dev, seed=1020 + rank)
    # The pool is scored with a FROZEN copy of the model whose classification head is "trained-like" (SURVEY 8d C3): the training phase
    # of the bench fits random labels, which flattens any synthetic confidence within a few SGD steps, and random-init confidence is
    # ~0.05 everywhere -- either way every image would score 0 and the HUA sampler would idle.  Same architecture, same kernels.
    import copy
    pool_model = copy.deepcopy(model)
    cal_k, cal_frac = calibrate_head(pool_model, pool['img'])
    broadcast_model(pool_model)               # every rank scores with rank 0's calibrated head
    score_kw = dict(return_loss=False, rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS',
                    uPool2='objectSum_scaleMax_classSum', scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
    do_train, do_score = 'train' in args.mode, 'score' in args.mode
    have_scoring = True
    try:
        import aod_meh_hua_amd.scoring  # noqa: F401
    except ImportError:
        have_scoring = False
        do_score = False

    from aod_meh_hua_amd.graphs import GraphedScore, GraphedTrainStep
    use_graph = not args.no_graph
    gstep = GraphedTrainStep(model, opt, opt_L, grad_sync=gsync if world > 1 else None, Labeled=True, Pseudo=False)
    gscore = GraphedScore(pool_model, **{k: v for k, v in score_kw.items() if k != 'return_loss'}) if have_scoring else None
    data_dev = dict(data, gt_bboxes=[b.to(dev) for b in data['gt_bboxes']], gt_labels=[l.to(dev) for l in data['gt_labels']])

    state = dict(graph_ok=use_graph)

    def step(it=0, do_train=do_train, do_score=do_score, graph=None):
        """One bench step.  graph=True replays the captured HIP graphs (same kernels, same work); graph=False enqueues from Python."""
        graph = state['graph_ok'] if graph is None else graph
        if graph:
            if do_train:
                gstep(data)
            if do_score:
                ids = torch.arange(B, device=dev) + (it * world + rank) * B
                _, unc = gscore(pool['img'], pool['img_metas'], ids)
                if world > 1:
                    gather_scores(unc, B * world)
            return
        if do_train:
            model.train()
            out, head_out, feat_out, prev = model.train_step(data_dev, Labeled=True, Pseudo=False)
            opt.zero_grad()
            out['loss'].backward()
            pending = gsync.start(opt.param_groups[0]['params'])       # overlaps the MEH step (disjoint parameters, detached inputs)
            lossL = model.train_step_L(prev, head_out, feat_out)
            opt_L.zero_grad()
            lossL['loss'].backward()
            pending.wait()
            opt.step()
            gsync.all_reduce_grads(opt_L.param_groups[0]['params'])
            opt_L.step()
        if do_score:
            pool_model.eval()
            with torch.no_grad():
                ids = torch.arange(B, device=dev) + (it * world + rank) * B
                _, unc = pool_model(img=[pool['img']], img_metas=[pool['img_metas']], image_ids=ids, **score_kw)
                unc = torch.as_tensor(unc, device=dev, dtype=torch.float32)
                if world > 1:
                    gather_scores(unc, B * world)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph:        # capture up front; if the runtime refuses (it must not take the bench down), fall back to the eager path on ALL ranks
        ok = 1
        try:
            step(0)
        except Exception as e:      # noqa: BLE001
            ok = 0
            print(f'[bench] HIP-graph capture failed on rank {rank}: {type(e).__name__}: {e}; falling back to eager launches', file=sys.stderr)
        if world > 1:
            import torch.distributed as dist
            t_ok = torch.tensor([ok], device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok = int(t_ok)
        state['graph_ok'] = use_graph = bool(ok)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    phases = int(do_train) + int(do_score)
    imgs_per_step = B * world * phases
    value = imgs_per_step * args.steps / dt

    # ---- per-phase rates (outside the timed region; SURVEY 8d reports train / score separately beside the combined rate)
    phase = {}
    for name, tr_, sc_ in (('train', True, False), ('score', False, True)):
        if (tr_ and not do_train) or (sc_ and not do_score):
            continue
        k = max(3, min(args.steps, 10))
        barrier()
        t1 = time.perf_counter()
        for i in range(k):
            step(args.warmup + args.steps + i, tr_, sc_)
        barrier()
        d = time.perf_counter() - t1
        phase[name + '_img_per_s'] = round(B * world * k / d, 1)
        phase[name + '_ms_per_batch'] = round(d / k * 1e3, 3)

    # ---- roofline of the dominant kernel: one extra instrumented step, HIP events around every conv launch
    roof = None
    barrier()
    if rank == 0:
        ho.PROFILE = []
    step(args.warmup + args.steps, graph=False)              # EVERY rank (the step contains collectives); HIP events need the eager path
    barrier()
    if rank == 0:
        agg = {}
        for kind, shape, flops, e0, e1 in ho.PROFILE:
            a = agg.setdefault(kind, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += flops
        if args.shapes:
            by = {}
            for kind, shape, flops, e0, e1 in ho.PROFILE:
                a = by.setdefault((kind,) + tuple(shape), [0, 0.0, flops])
                a[0] += 1
                a[1] += e0.elapsed_time(e1) * 1e3
            with open(args.shapes, 'w') as f:
                f.write('kind      M       N     K   RS st   n   us_each   TFLOP/s   GB/s(act in+out)\n')
                for (kind, m, n, k, rs, st), (cnt, us, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                    c = k // rs
                    m_in = m * st * st if kind == 'fwd' else m
                    byts = 2.0 * (m_in * (c if kind != 'dgrad' else n) + m * (n if kind != 'dgrad' else c)) if kind != 'wgrad' else 2.0 * (m_in * c + m * n)
                    f.write(f'{kind:6s}{m:8d}{n:6d}{k:6d}{rs:4d}{st:3d}{cnt:4d}{us / cnt:10.1f}{fl / (us / cnt) / 1e6:10.1f}{byts / (us / cnt) / 1e3:10.1f}   total {us:8.1f}\n')
        ho.PROFILE = None
        kind = max(agg, key=lambda k: agg[k][1])
        n, tsec, fl = agg[kind]
        # HBM-side bytes per launch of the dominant kernel: rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ pass (tools/dbg/pmc_bench.sh), corrected as
        # MI355X_MICROARCH.md prescribes (128 B per non-32B read request on gfx950); measured offline, committed under profiles/
        traffic = None
        try:
            pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01n_pmc_hbm_traffic_per_launch.json')))
            # (launch-weighted over the instances of the kernel: 4- / 8-wave forms, epilogue-operand variants of the 128 x 128 tile)
            pref = 'void conv_wgrad_kernel' if kind == 'wgrad' else 'void conv_igemm_kernel<128, 128'
            ks = [k for k in pm if k.startswith(pref)]
            nl = sum(pm[k]['launches'] for k in ks)
            traffic = round(sum((pm[k]['read_MB_per_launch'] + pm[k]['write_MB_per_launch']) * pm[k]['launches'] for k in ks) / nl * 1e6)
        except Exception:      # noqa: BLE001
            pass
        roof = dict(bound='mfma', kernel={'fwd': 'conv_igemm_kernel (forward)', 'dgrad': 'conv_igemm_kernel (dgrad)', 'wgrad': 'conv_wgrad_kernel'}[kind],
                    achieved=round(fl / tsec / 1e12, 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s', frac=round(fl / tsec / 1e12 / PEAK_BF16_TFLOPS, 4),
                    traffic=traffic, launches_per_step=n, avg_launch_us=round(tsec / n * 1e6, 2),
                    all={k: dict(launches=v[0], ms=round(v[1] * 1e3, 3), tflops=round(v[2] / v[1] / 1e12, 1)) for k, v in agg.items()})

    hua = None
    if do_score:
        hua = hua_stats(pool_model, pool, score_kw, dev)          # every rank (same launches); rank 0 reports its own batch
        hua.update(head_scale=round(cal_k, 3), fg_anchor_frac_at_calibration=round(cal_frac, 5))
        assert hua['pairs_per_img'] > 0, 'degenerate HUA phase: no (candidate, object) pair in the scoring batch'

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, do_train, do_score and have_scoring)

    if rank == 0:
        line = dict(metric='images/sec train+HUA-score, RetinaNet-R50 VOC 512^2', value=round(value, 2), unit='images/sec',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='bf16', data='synthetic',
                    config=dict(workload=f'RetinaNet-R50-FPN + MEH/HUA, synthetic VOC {H}x{H}, bs={B}/GPU: '
                                         + ' + '.join((['train iteration (main fwd/bwd/SGD + MEH fwd/bwd/SGD)'] if do_train else [])
                                                      + (['HUA scoring pass'] if do_score else [])),
                                global_batch=B * world, image_size=H, num_classes=20, anchors_per_image=49104 if H == 512 else None,
                                parallelism=f'dp{world}', phases=args.mode if (do_score or not have_scoring) else 'train',
                                launch='hip-graph replay' if use_graph else 'eager'),
                    phase_rates=phase, hua=hua, roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(args, do_train, do_score):
    """CPU oracle (port of the reference's CPU path) on a bounded sample: B=2 (the reference's samples_per_gpu,
    Config_RetinaNet.py:127) at the bench resolution, all host cores."""
    from oracle import model as om
    # torch's CPU conv/backward kernels stop scaling (and thrash) far below the 256 host threads of the GPU box:
    # use 16 threads -- 8x the reference's own torch.set_num_threads(2) (tools/train_RetinaNet.py:77) -- and say so.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    H = args.size
    sd = om.seeded_state_dict()
    train_keys = [k for k, v in sd.items() if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.'))]
    for k in train_keys:
        sd[k].requires_grad_(True)
    g = torch.Generator().manual_seed(20)
    B = 2
    img = torch.randn(B, 3, H, H, generator=g)
    gtb = [torch.tensor([[H * .1, H * .2, H * .6, H * .7]]), torch.tensor([[H * .3, H * .3, H * .8, H * .9], [H * .05, H * .05, H * .3, H * .4]])]
    gtl = [torch.tensor([3]), torch.tensor([7, 1])]
    bufs, bufs_L = {}, {}
    meh = [k for k in train_keys if 'retina_L' in k or 'L_convs' in k]
    main = [k for k in train_keys if k not in meh]

    def it():
        n = 0
        if do_train:
            o = om.train_step(sd, img, gtb, gtl)
            for k in train_keys:
                sd[k].grad = None
            o['loss'].backward()
            with torch.no_grad():
                om.sgd_step({k: sd[k] for k in main}, {k: sd[k].grad for k in main}, bufs)
            oL = om.train_step_L(sd, o['feats'], o['loss_noR'], o['targets'])
            for k in train_keys:
                sd[k].grad = None
            oL['loss'].backward()
            with torch.no_grad():
                om.sgd_step({k: sd[k] for k in meh}, {k: sd[k].grad for k in meh}, bufs_L)
            n += B
        if do_score:
            with torch.no_grad():
                om.score_images(sd, img, sampler='torch')
            n += B
        return n
    t0 = time.perf_counter()
    n, iters = it(), 1                      # first iteration doubles as warm-up if it already exhausts the budget
    dt = time.perf_counter() - t0
    if dt < args.cpu_seconds:
        t0, n, iters = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < args.cpu_seconds:
            n += it()
            iters += 1
        dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 3), unit='images/sec', cores=cores, kind='port',
                sample=f'{iters} iterations of the same step at B={B}, {H}x{H}, fp32 torch CPU ops, {cores} threads (oracle/model.py)')


if __name__ == '__main__':
    main()
