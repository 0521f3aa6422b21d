#!/usr/bin/env python
"""Benchmark of the MEH/HUA hot path on MI355X (contract: prompt section 'bench.py').

Default workload (BASELINE.json configs[1]): RetinaNet-R50-FPN + MEH/HUA, synthetic VOC 512x512, 20 classes, 16 images
per GPU.  One "step" = one AL work unit over one batch of synthetic images resident in HBM: the full training iteration of
MyEpochBasedRunnerLambda.run_iter (main forward + backward + SGD, then MEH forward + backward + SGD) on 16 images PLUS the HUA scoring
pass over 16 images of the same shape (forward + MEH forward + top-k + NMS + Dirichlet sampling + aggregation -> one score per image),
i.e. the metric "images/sec train + HUA-score".  value = SURVEY 8(d)'s AL-cycle rate N / (T_train + T_score) = 16 * GPUs * steps / time
(images through BOTH phases per second); `phase_rates` holds the two phases separately (>= 50 iterations each).

Arithmetic: --precision bf16x3 (default) is the REFERENCE-PRECISION mode -- the reference is fp32 end to end, so the headline is measured
with fp32-grade products (bf16 head/tail pairs, three MFMAs per product, aod_meh_hua_amd/functional.py set_precision, csrc/conv.hip "X3"); --precision bf16 is the fast
mode with plain bf16 operands.  The line carries the other mode's figures under precision.other_mode (N = 1).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)
    python bench.py --config r101coco                        BASELINE configs[4] at N = 1: R101, 80 classes, 8 x 800x1344 per GPU
    python bench.py --mode pool --pool 10000                 BASELINE configs[3]: the real pool loop over on-device Philox images

Prints ONE JSON line on rank 0.  `roofline` is measured live (HIP events on the launch stream around every conv launch of one extra
instrumented step); `cpu_baseline` times the CPU oracle (a faithful port of the reference's CPU path, oracle/model.py) on a bounded
sample on rank 0 at N == 1 only.
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec (6.29 TB/s measured with a float4 copy)

CONFIGS = {
    'voc512': dict(depth=50, classes=20, H=512, W=512, batch=16, name='RetinaNet-R50-FPN + MEH/HUA, synthetic VOC 512x512',
                   metric='images/sec train+HUA-score, RetinaNet-R50 VOC 512^2'),
    'r101coco': dict(depth=101, classes=80, H=800, W=1344, batch=8, name='RetinaNet-R101-FPN + MEH/HUA, synthetic COCO 800x1344 (80 classes)',
                     metric='images/sec train+HUA-score, RetinaNet-R101 COCO 800x1344'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='voc512', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--size', type=int, default=None, help='square image side (overrides the config)')
    ap.add_argument('--mode', default='train+score', choices=['train', 'score', 'train+score', 'pool'])
    ap.add_argument('--pool', type=int, default=10000, help='--mode pool: number of pool images (all ranks together)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'bf16'],
                    help='arithmetic of the conv stack: bf16x3 = reference precision (fp32-grade products from bf16 head/tail pairs; the default, the '
                         'reference is fp32), bf16 = plain bf16 operands (faster, residuals ~1e-2)')
    ap.add_argument('--deterministic', action='store_true', help='ordered bias / BN-shift column sums (functional.set_deterministic): bit-repeatable runs, ~1 %% slower')
    ap.add_argument('--phase-iters', type=int, default=50, help='iterations of each per-phase rate (SURVEY 8d: >= 50)')
    ap.add_argument('--no-precision-check', action='store_true', help='skip the short run in the OTHER precision mode after the timed region (profiling runs)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--shapes', default=None, help='write the per-shape conv breakdown of the instrumented step to this file')
    ap.add_argument('--serial-scores', action='store_true', help='timed loop: read (and, for N > 1, all-gather) the scores after every step instead of once after the loop')
    ap.add_argument('--rotate', type=int, default=4, help='distinct resident train batches / pool batches the timed loop cycles through (1: one batch, '
                    'written straight into the graphs\' static input buffers)')
    ap.add_argument('--no-graph', action='store_true', help='enqueue every kernel from Python instead of replaying HIP graphs')
    return ap.parse_args()


def synth_batch(B, H, W, device, seed, classes=20):
    """img ~ N(0,1); G ~ U{1..5}; w,h ~ U(32,384) (x H/512) clipped inside; labels ~ U{0..classes-1}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, H, W, generator=g)
    boxes, labels = [], []
    for _ in range(B):
        G = int(torch.randint(1, 6, (1,), generator=g))
        wh = torch.rand(G, 2, generator=g) * (384 - 32) * (H / 512.0) + 32 * (H / 512.0)
        xy = torch.rand(G, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh).clamp(min=0)
        x2y2 = torch.minimum(xy + wh, torch.tensor([float(W), float(H)]))
        boxes.append(torch.cat([xy, x2y2], 1))      # ground truth stays on the host like a data loader's
        labels.append(torch.randint(0, classes, (G,), generator=g))
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=np.ones(4, np.float32), flip=False)
             for _ in range(B)]
    return dict(img=img.to(device), img_metas=metas, gt_bboxes=boxes, gt_labels=labels)


def build_model(device, cd, seed=20):
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')        # no network: random-init weights of the named architecture
    cfg.model.backbone.depth = cd['depth']
    cfg.model.bbox_head.num_classes = cd['classes']
    cfg.model.bbox_head.loss_cls.num_classes = cd['classes']
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
        cls, _ = head.forward(model.extract_feat(img[:min(4, img.shape[0])]))
        x = torch.cat([nhwc_view(c.float(), head.cls_out_channels) for c in cls], 1)
        frac = lambda k: float((torch.softmax(x * k, -1).amax(-1) > fg_thr).float().mean())
        lo, hi = 1.0, 2.0
        while frac(hi) < target_frac and hi < 1e6:
            lo, hi = hi, hi * 2
        for _ in range(24):
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
    out = dict(bound='valu', pairs_per_img=round(pairs / B, 1), objects_per_img=round(float(nobj.float().mean()), 1),
               nonzero_scores=int((unc > 0).sum()), hua_us_per_batch=round(us, 1), us_per_pair=round(us / max(pairs, 1), 4),
               gamma_variates_per_s=round(pairs * 500 * nd / (us * 1e-6), 0) if pairs else 0.0, samples_per_pair=500, dirichlet_columns=nd)
    pm = pmc_summary('pmc_hua.json')        # separate rocprofv3 --pmc pass of THIS build (tools/profile/pmc_passes.sh), else null
    out.update(valu_util=pm.get('valu_util') if pm else None, valu_issue_frac_of_peak=pm.get('valu_issue_frac_of_peak') if pm else None,
               valu_util_source=f'profiles/pmc_hua.json @ kernels {pm["kernels_sha16"]}' if pm else None)
    return out


def pmc_summary(name):
    """A committed PMC summary (profiles/<name>, written by tools/profile/pmc_passes.sh: rocprofv3 --pmc cannot run inside a timed bench) is
    merged into the line ONLY when it was measured on this very build: its `kernels_sha16` must equal the digest of the kernel sources.
    Otherwise the field is null -- a stale counter in a live line is worse than none."""
    try:
        from aod_meh_hua_amd.build import source_digest
        pm = json.load(open(os.path.join(ROOT, 'profiles', name)))
        return pm if pm.get('kernels_sha16') == source_digest() else None
    except Exception:      # noqa: BLE001
        return None


def make_optimizers(model, cfg):
    """apis/train_Lambda.py:54-61: main SGD without the MEH parameters + optimizer_L over the MEH parameters."""
    from aod_meh_hua_amd.optim import FusedSGD
    o = cfg.optimizer
    meh = [p for n, p in model.named_parameters() if ('retina_L' in n or 'L_convs' in n)]
    ids = {id(p) for p in meh}
    main = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    return (FusedSGD(main, lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay),
            FusedSGD(meh, lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay))


SCORE_KW = dict(return_loss=False, rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum',
                scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per GPU, the way the
    reference's launcher does: tools/train_RetinaNet.py:119-121 init_dist(args.launcher, ...)).  Runs BEFORE anything touches the GPU and as
    a CHILD process (a process that has initialised HIP must never exec another program); returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')).returncode


def measured_peaks(dev):
    """the vendor peaks re-measured on THIS box inside the run -- a library bf16 GEMM (hipBLASLt through torch.matmul, 8192^3,
    random operands) and a stream copy (1 GiB, torch's copy kernel) -- about 2 s together.  Calibration only: nothing of the product path
    goes through either."""
    try:
        g = torch.Generator(device=dev).manual_seed(1)
        n = 8192
        a = torch.randn(n, n, device=dev, generator=g).bfloat16()
        b = torch.randn(n, n, device=dev, generator=g).bfloat16()
        for _ in range(3):
            torch.matmul(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record()
        for _ in range(reps):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        gemm = 2.0 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
        del a, b
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_(generator=g)
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        e0.record()
        for _ in range(20):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy = 2.0 * src.numel() * 4 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        # the shader clock the chip holds under matrix load (csrc/probe.hip): ~0.4 s of back-to-back MFMA launches, median over workgroups
        clock = mfma_tf = None
        try:
            from aod_meh_hua_amd import hipops as ho_
            nwg, iters = 512, 20000
            outp = torch.zeros(2 * nwg, dtype=torch.int64, device=dev)
            sink = torch.zeros(1, device=dev)
            for _ in range(6):
                ho_.call('aod_mfma_clock_probe', iters, nwg, ho_.ptr(outp), ho_.ptr(sink), ho_.stream())
            e0.record()
            ho_.call('aod_mfma_clock_probe', iters, nwg, ho_.ptr(outp), ho_.ptr(sink), ho_.stream())
            e1.record()
            torch.cuda.synchronize()
            o = outp.cpu().numpy().reshape(nwg, 2).astype(np.float64)
            clock = float(np.median(o[:, 0] / np.maximum(o[:, 1], 1) * 0.1))          # cycles per 10-ns tick -> GHz
            mfma_tf = 2.0 * 16 * 16 * 32 * 16 * iters * nwg * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        except Exception:      # noqa: BLE001
            pass
        return dict(gemm_bf16_tflops=round(gemm, 1), gemm='torch.matmul (hipBLASLt) 8192^3 bf16, random operands', stream_copy_GBs=round(copy, 1),
                    stream_copy='1 GiB fp32 device-to-device copy, read + write bytes',
                    mfma_loop_clock_ghz=None if clock is None else round(clock, 3), mfma_loop_tflops=None if mfma_tf is None else round(mfma_tf, 1),
                    mfma_loop='register-resident v_mfma_f32_16x16x32_bf16 loop on random operands, every SIMD busy (csrc/probe.hip): the clock the chip '
                              'holds under matrix load and the rate the pipe delivers at it (the 2.5 PFLOP/s peak is quoted at 2.4 GHz)')
    except Exception as e:      # noqa: BLE001
        return dict(error=f'{type(e).__name__}: {e}'[:200])


PRECISIONS = {
    # mode -> (dtype string of the line, MFMA instructions per algorithmic product, description)
    'bf16x3': ('bf16x3', 3, 'reference precision: activations / gradients / filters as bf16 head + tail pairs, xh*wh + xl*wh + xh*wl on the bf16 '
                            'MFMA into fp32 accumulators (16-bit significands, fp32-grade results: golden train step and scoring at 1e-4); '
                            'fp32 losses / geometry / scoring / optimizer'),
    'bf16': ('bf16', 1, 'bf16 x bf16 -> fp32 MFMA convolutions (8-bit significands: narrower than the fp32 reference, residuals ~1e-2); fp32 '
                        'losses / geometry / scoring / optimizer'),
}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    cd = dict(CONFIGS[args.config])
    if args.size:
        cd.update(H=args.size, W=args.size, name=cd['name'] + f' [resized to {args.size}x{args.size}]')
    B, H, W = args.batch or cd['batch'], cd['H'], cd['W']
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # debugging aid only: AOD_BENCH_ONE_GPU=1 runs every rank on device 0 over gloo (functional check of the multi-rank path on a
    # single-GPU box); the driver's multi-GPU runs use one GPU per rank over RCCL
    one_gpu = os.environ.get('AOD_BENCH_ONE_GPU') == '1'
    local = 0 if one_gpu else local
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (a line must not measure fewer ranks than it names)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if args.deterministic:
        from aod_meh_hua_amd import functional as _AF
        _AF.set_deterministic(True)
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    comm = dict(backend=None, ranks=1)
    if world > 1:
        # the number of ranks a COLLECTIVE actually spanned (one all-reduce of ones over the freshly initialised group), not the launcher's claim
        # -- and WHICH ranks: a presence vector summed over the group names the absent ones in the failure message (before any timing starts)
        present = torch.zeros(max(args.gpus, dist.get_world_size()), device=dev)
        present[rank] = 1
        dist.all_reduce(present)
        torch.cuda.synchronize()
        seen = [i for i, v in enumerate(present.tolist()) if v > 0]
        comm = dict(backend='gloo (debug: all ranks on one GPU)' if one_gpu else 'nccl (RCCL)', ranks=len(seen))
        missing = sorted(set(range(args.gpus)) - set(seen))
        if comm['ranks'] != args.gpus or dist.get_world_size() != args.gpus or missing or max(present.tolist()) > 1:
            raise RuntimeError(f'bench.py --gpus {args.gpus}: the {comm["backend"]} group spans ranks {seen} (world size {dist.get_world_size()}); '
                               f'missing ranks: {missing or "none"}; ranks counted twice: {[i for i, v in enumerate(present.tolist()) if v > 1] or "none"} '
                               '-- launch one process per GPU (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)')
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd.parallel import GradSync, broadcast_model, gather_scores
    import aod_meh_hua_amd.scoring  # noqa: F401      (the HIP scoring pass is part of the product: no fallback)
    AF.set_precision(args.precision)
    model, cfg = build_model(dev, cd)
    broadcast_model(model)
    # The pool is scored with a FROZEN copy of the model whose classification head is "trained-like" (SURVEY 8d C3): the training phase
    # of the bench fits random labels, which flattens any synthetic confidence within a few SGD steps, and random-init confidence is
    # ~0.05 everywhere -- either way every image would score 0 and the HUA sampler would idle.  Same architecture, same kernels.
    R = max(1, args.rotate)
    # (VERDICT r5 "bench realism": the loop cycles through R different batches -- other ground-truth counts / boxes, other confidence maps -- so
    # that assignment, loss reductions and the HUA pair count do not see one pattern for the whole run)
    pools = [synth_batch(B, H, W, dev, seed=1020 + rank + 1000 * k, classes=cd['classes']) for k in range(R)]
    pool = pools[0]
    pool_model = copy.deepcopy(model)
    cal_k, cal_frac = calibrate_head(pool_model, pool['img'])
    broadcast_model(pool_model)               # every rank scores with rank 0's calibrated head

    if args.mode == 'pool':
        return pool_mode(args, cd, pool_model, dev, rank, world, B, H, W, cal_k, cal_frac, comm)

    opt, opt_L = make_optimizers(model, cfg)
    gsync = GradSync()
    datas = [synth_batch(B, H, W, dev, seed=20 + rank + 1000 * k, classes=cd['classes']) for k in range(R)]
    data = datas[0]
    do_train, do_score = 'train' in args.mode, 'score' in args.mode
    phases = int(do_train) + int(do_score)

    from aod_meh_hua_amd.graphs import GraphedScore, GraphedTrainStep
    data_devs = [dict(d_, gt_bboxes=[b.to(dev) for b in d_['gt_bboxes']], gt_labels=[l.to(dev) for l in d_['gt_labels']]) for d_ in datas]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup, use_graph, phase_iters):
        """Everything the line says about ONE precision mode: the timed region (warm-up, barrier, `steps` steps, barrier; max over ranks), the
        per-phase rates and one instrumented eager step for the roofline blocks."""
        AF.set_precision(precision)
        gstep = GraphedTrainStep(model, opt, opt_L, grad_sync=gsync if world > 1 else None, Labeled=True, Pseudo=False)
        gscore = GraphedScore(pool_model, **{k: v for k, v in SCORE_KW.items() if k != 'return_loss'})
        state = dict(graph_ok=use_graph, filled=set())

        def id_table(it, tab={}):
            """global image ids of scoring batch `it` as a slice of one resident arange (what the pool loop hands over: a slice of its id
            vector) -- `arange + offset` per step were two 5-us launches in front of every scoring batch"""
            lo = (it * world + rank) * B
            t = tab.get('t')
            if t is None or lo + B > t.numel():
                t = tab['t'] = torch.arange(max(2 * (lo + B), 1 << 16), device=dev)
            return t[lo:lo + B]

        def step(it=0, do_train=do_train, do_score=do_score, graph=None, defer=False):
            """One bench step.  graph=True replays the captured HIP graphs (same kernels, same work); graph=False enqueues from Python.
            defer=True (score-only loops): like the pool loop of apis/test.py, the scores are not read before the loop's end, so the selection
            half of batch k (second stream) runs beside the conv half of batch k + 1 (graphs.GraphedScore)."""
            graph = state['graph_ok'] if graph is None else graph
            k_ = it % R
            if graph:
                if do_train:
                    # R > 1: batch k_ is copied into the graph's static input buffer (one 50 MB device copy) and its ground truth packed and
                    # uploaded, as for a loader's batch; R == 1: `data` IS the static buffer (see below)
                    gstep(data if R == 1 else datas[k_])
                if do_score:
                    ids = id_table(it)
                    # R == 1: the synthetic pool batch is resident in HBM (contract): it sits in the static input buffer of EACH of the scoring
                    # graph's alternating slots -- where the on-device pool generator / a loader's H2D copy writes a real batch
                    # (graphs.static_image) -- so no 50 MB device-to-device copy rides in the step; R > 1: batch k_ is copied in per step
                    img = (pool if R == 1 else pools[k_])['img']
                    si = gscore.static_image(tuple(img.shape)) if R == 1 else None
                    if si is not None:
                        if si.data_ptr() not in state['filled']:
                            si.copy_(img)
                            state['filled'].add(si.data_ptr())
                        img = si
                    _, unc = gscore(img, pools[k_]['img_metas'], ids, defer=defer)
                    if world > 1 and not defer:
                        gather_scores(unc, B * world)
                    return unc
                return None
            if do_train:
                model.train()
                out, head_out, feat_out, prev = model.train_step(data_devs[k_], Labeled=True, Pseudo=False)
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
                    ids = id_table(it)
                    _, unc = pool_model(img=[pools[k_]['img']], img_metas=[pools[k_]['img_metas']], image_ids=ids, **SCORE_KW)
                    unc = torch.as_tensor(unc, device=dev, dtype=torch.float32)
                    if world > 1:
                        gather_scores(unc, B * world)

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
        if state['graph_ok']:
            # the synthetic batches are resident in HBM (contract); hand the graphs their OWN static input buffers as the batch -- what a loader
            # that fills graphs.static_image() directly does -- so that no 50 MB device-to-device copy per phase rides in the step
            nonlocal data, pool
            if R == 1 and do_train and getattr(gstep, 'cur', None) and tuple(gstep.cur['static']['img'].shape) == tuple(data['img'].shape):
                data = dict(data, img=gstep.cur['static']['img'])
        # The timed loop calls the scoring graph the way the product's pool loop does (apis/test.py single_gpu_uncertainty): scores are
        # deferred -- the selection half of batch k (<= 16 workgroups) runs on its own stream beside whatever follows --, read after the
        # loop (gscore.sync()) and all-gathered ONCE, all inside the timed region.  --serial-scores restores one read + one gather per step.
        deferred = state['graph_ok'] and do_score and not args.serial_scores

        def timed(n, first):
            uncs = []
            barrier()
            t0 = time.perf_counter()
            for i in range(n):
                uncs.append(step(first + i, defer=deferred))
            if deferred:
                gscore.sync()
                if world > 1:
                    gather_scores(torch.cat(uncs), B * world * n)
            barrier()
            return time.perf_counter() - t0

        for i in range(warmup):
            step(i)
        dt = timed(steps, warmup)
        serial_ms = None
        if deferred and world == 1:                   # the same steps with the scores read after every step, beside the line (not `value`)
            deferred = False
            serial_ms = round(timed(min(steps, 20), warmup + steps) / min(steps, 20) * 1e3, 3)
            deferred = True
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)

        # ---- per-phase rates (outside the timed region; SURVEY 8d: >= 50 iterations, train / score separately beside the combined rate)
        phase = {}
        for name, tr_, sc_ in (('train', True, False), ('score', False, True)):
            if (tr_ and not do_train) or (sc_ and not do_score):
                continue
            k = phase_iters
            barrier()
            t1 = time.perf_counter()
            us_ = [step(warmup + steps + i, tr_, sc_, defer=sc_ and not tr_) for i in range(k)]
            gscore.sync()
            if world > 1 and sc_ and not tr_ and state['graph_ok']:
                gather_scores(torch.cat(us_), B * world * k)
            barrier()
            d = time.perf_counter() - t1
            phase[name + '_img_per_s'] = round(B * world * k / d, 1)
            phase[name + '_ms_per_batch'] = round(d / k * 1e3, 3)
            phase['iterations'] = k
        if do_train and do_score:
            # AL-cycle rate N_total / (T_train + T_score) when the SAME N images are first trained on and then scored
            phase['al_cycle_img_per_s'] = round(B * world / ((phase['train_ms_per_batch'] + phase['score_ms_per_batch']) * 1e-3), 1)

        # ---- roofline of the dominant kernel: one extra instrumented step, HIP events around every conv launch
        roof = None
        barrier()
        if rank == 0:
            ho.PROFILE, ho.BYTES_PROFILE = [], []
        # The eager step is enqueued from Python far more slowly than the device executes it; an event pair around a launch would then span
        # the idle time until the launch arrives (~10 us per small kernel).  A device-side spin keeps the stream busy while the host enqueues
        # the whole step, so that the events are processed back to back and a pair brackets nothing but its kernel (the rocprofv3 averages
        # under profiles/ are the cross-check).
        torch.cuda._sleep(int(0.6 * 2.0e9))
        step(warmup + steps, graph=False)              # EVERY rank (the step contains collectives); HIP events need the eager path
        barrier()
        # ... twice, keeping the SHORTER time of every launch: a single instrumented step now and then contains one launch that took
        # milliseconds (2.45 ms for a 224-us dgrad in one of this round's runs, never in the three repeats): a box hiccup, not a kernel time
        first = None
        if rank == 0:
            first = ([(k_, sh, fl, e0.elapsed_time(e1), sc) for k_, sh, fl, e0, e1, sc in ho.PROFILE], [(n_, nb, e0.elapsed_time(e1)) for n_, nb, e0, e1 in ho.BYTES_PROFILE])
            ho.PROFILE, ho.BYTES_PROFILE = [], []
        torch.cuda._sleep(int(0.6 * 2.0e9))
        step(warmup + steps, graph=False)
        barrier()
        if rank == 0:
            second = ([(k_, sh, fl, e0.elapsed_time(e1), sc) for k_, sh, fl, e0, e1, sc in ho.PROFILE], [(n_, nb, e0.elapsed_time(e1)) for n_, nb, e0, e1 in ho.BYTES_PROFILE])

            class _Ev:          # (the aggregation below reads e0.elapsed_time(e1))
                def __init__(self, ms):
                    self.ms = ms

                def elapsed_time(self, other):
                    return other.ms

            def _merge(a, b, ti):
                if len(a) != len(b) or any(x[:ti] != y[:ti] for x, y in zip(a, b)):
                    return b          # (the two steps did not issue the same launches: keep the later one)
                return [x[:ti] + (min(x[ti], y[ti]),) + x[ti + 1:] for x, y in zip(a, b)]
            conv_m, bytes_m = _merge(first[0], second[0], 3), _merge(first[1], second[1], 2)
            ho.PROFILE = [(k_, sh, fl, _Ev(0.0), _Ev(ms), sc) for k_, sh, fl, ms, sc in conv_m]
            ho.BYTES_PROFILE = [(n_, nb, _Ev(0.0), _Ev(ms)) for n_, nb, ms in bytes_m]
        if rank == 0:
            mfma_per_product = PRECISIONS[precision][1]
            peak = PEAK_BF16_TFLOPS / mfma_per_product      # algorithmic (fp32-grade) FLOP/s the matrix pipe can deliver in this mode
            agg, parts = {}, {}
            for kind, shape, flops, e0, e1, scope in ho.PROFILE:
                t_ = e0.elapsed_time(e1) * 1e-3
                a = agg.setdefault(kind, [0, 0.0, 0.0])
                a[0] += 1; a[1] += t_; a[2] += flops
                part = 'backbone_fpn' if scope in ('backbone', 'neck') else 'heads'
                q = parts.setdefault(part, {}).setdefault(kind, [0, 0.0, 0.0])
                q[0] += 1; q[1] += t_; q[2] += flops
            if args.shapes and precision == args.precision:
                by = {}
                for kind, shape, flops, e0, e1, scope in ho.PROFILE:
                    a = by.setdefault((kind, scope) + tuple(shape), [0, 0.0, flops])
                    a[0] += 1
                    a[1] += e0.elapsed_time(e1) * 1e3
                with open(args.shapes, 'w') as f:
                    f.write(f'# precision {precision}; TFLOP/s = algorithmic (2*M*R*S*Cin*Cout) / time\n')
                    f.write('kind   part          M       N     K   RS st   n   us_each   TFLOP/s   GB/s(act in+out)\n')
                    for (kind, scope, m, n, k, rs, st), (cnt, us, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                        c = k // rs
                        m_in = m * st * st if kind == 'fwd' else m
                        byts = 2.0 * (m_in * (c if kind != 'dgrad' else n) + m * (n if kind != 'dgrad' else c)) if kind != 'wgrad' else 2.0 * (m_in * c + m * n)
                        f.write(f'{kind:6s} {scope:9s}{m:8d}{n:6d}{k:6d}{rs:4d}{st:3d}{cnt:4d}{us / cnt:10.1f}{fl / (us / cnt) / 1e6:10.1f}{byts / (us / cnt) / 1e3:10.1f}   total {us:8.1f}\n')
            sec = {}
            for name, nbytes, e0, e1 in ho.BYTES_PROFILE:
                a = sec.setdefault(name, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += nbytes
            ho.PROFILE = ho.BYTES_PROFILE = None
            kind = max(agg, key=lambda k: agg[k][1])
            n, tsec, fl = agg[kind]
            # HBM-side bytes per launch of the dominant kernel class: separate rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ pass of THIS build
            # (tools/profile/pmc_passes.sh; corrected as MI355X_MICROARCH.md prescribes: 128 B per non-32B read request on gfx950), null when the
            # committed summary belongs to another build or another precision mode
            traffic, traffic_src = None, None
            pm = pmc_summary('pmc_traffic.json' if precision == 'bf16x3' else 'pmc_traffic_bf16.json')
            if pm and pm.get('precision', 'bf16') == precision:
                # (launch-weighted over the instances of the kernel: 4- / 8-wave forms, epilogue-operand variants, fused forward kernels)
                pref = ('conv_wgrad_kernel',) if kind == 'wgrad' else ('conv_igemm_kernel', 'bottleneck', 'stem_pool_kernel', 'pw_gemm_kernel', 'pred_conv')
                ks = [k for k in pm['kernels'] if any(q in k for q in pref)]
                nl = sum(pm['kernels'][k]['launches'] for k in ks)
                if nl:
                    traffic = round(sum((pm['kernels'][k]['read_MB_per_launch'] + pm['kernels'][k]['write_MB_per_launch']) * pm['kernels'][k]['launches'] for k in ks) / nl * 1e6)
                    traffic_src = f'profiles/pmc_traffic{"" if precision == "bf16x3" else "_bf16"}.json @ kernels {pm["kernels_sha16"]}'
            rate = lambda v: dict(launches=v[0], ms=round(v[1] * 1e3, 3), tflop=round(v[2] / 1e12, 3), tflops=round(v[2] / v[1] / 1e12, 1),
                                  frac=round(v[2] / v[1] / 1e12 / peak, 4))
            tot = lambda d: [sum(v[0] for v in d.values()), sum(v[1] for v in d.values()), sum(v[2] for v in d.values())]
            roof = dict(bound='mfma', kernel={'fwd': 'conv_igemm_kernel (forward)', 'dgrad': 'conv_igemm_kernel (dgrad)', 'wgrad': 'conv_wgrad_kernel'}[kind],
                        achieved=round(fl / tsec / 1e12, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(fl / tsec / 1e12 / peak, 4),
                        peak_rule=(f'{PEAK_BF16_TFLOPS:.0f} TFLOP/s dense bf16 MFMA / {mfma_per_product} MFMA per algorithmic product in the {precision} mode'
                                   + (' (the fp32 matrix pipe itself peaks at 157 TFLOP/s)' if mfma_per_product > 1 else '')),
                        mfma_issued_tflops=round(fl / tsec / 1e12 * mfma_per_product, 1),
                        traffic=traffic, traffic_source=traffic_src, launches_per_step=n, avg_launch_us=round(tsec / n * 1e6, 2),
                        flops_rule='algorithmic: 2*M*R*S*Cin*Cout of the reference layer (channel pads of the stem / prediction convs excluded; one '
                                   'multiply-add per product whatever the mode issues)',
                        all={k: rate(v) for k, v in agg.items()},
                        # north_star's target stack: every conv launch of the backbone (stem included) and the neck, forward alone and with its backward
                        backbone_fpn={**{k: rate(v) for k, v in parts.get('backbone_fpn', {}).items()}, 'total': rate(tot(parts.get('backbone_fpn', {'-': [0, 1e-30, 0.0]})))},
                        heads={**{k: rate(v) for k, v in parts.get('heads', {}).items()}, 'total': rate(tot(parts.get('heads', {'-': [0, 1e-30, 0.0]})))},
                        # the HBM-bound row kernels of the same step: algorithmic bytes / summed launch time (HIP events), fraction of 8 TB/s
                        secondary={k: dict(bound='hbm', launches=v[0], us=round(v[1] * 1e6, 1), achieved=round(v[2] / v[1] / 1e9, 1), peak=PEAK_HBM_GBS,
                                           unit='GB/s', frac=round(v[2] / v[1] / 1e9 / PEAK_HBM_GBS, 4)) for k, v in sec.items() if v[1] > 0})
        del gstep, gscore
        return dict(dt=dt, steps=steps, phase=phase, roof=roof, use_graph=use_graph, deferred=deferred, serial_ms=serial_ms)

    main_m = measure(args.precision, args.steps, args.warmup, not args.no_graph, args.phase_iters)
    dt, use_graph = main_m['dt'], main_m['use_graph']
    # the combined figure is the AL-cycle rate N / (T_train + T_score) -- a step trains on B images per GPU and scores B (other)
    # images per GPU, i.e. B images per GPU pass through BOTH phases per step
    value = B * world * args.steps / dt

    hua = None
    if do_score:
        hua = hua_stats(pool_model, pool, SCORE_KW, dev)          # every rank (same launches); rank 0 reports its own batch
        hua.update(head_scale=round(cal_k, 3), fg_anchor_frac_at_calibration=round(cal_frac, 5))
        assert hua['pairs_per_img'] > 0, 'degenerate HUA phase: no (candidate, object) pair in the scoring batch'

    # ---- the other precision mode, same workload, same run (N = 1 only: a figure beside the line, not part of `value`)
    other = None
    if world == 1 and not args.no_precision_check:
        om_ = 'bf16' if args.precision == 'bf16x3' else 'bf16x3'
        try:
            m2 = measure(om_, 20, 3, not args.no_graph, 20)
            r2 = m2['roof'] or {}
            other = dict(mode=om_, arithmetic=PRECISIONS[om_][2], value=round(B * 20 / m2['dt'], 2), ms_per_step=round(m2['dt'] / 20 * 1e3, 3),
                         phase_rates=m2['phase'], launch='hip-graph replay' if m2['use_graph'] else 'eager',
                         roofline={k: r2.get(k) for k in ('kernel', 'achieved', 'peak', 'frac', 'peak_rule', 'all', 'backbone_fpn', 'heads')})
        except Exception as e:      # noqa: BLE001
            other = dict(mode=om_, value=None, error=f'{type(e).__name__}: {e}'[:300])
        finally:
            AF.set_precision(args.precision)
    prec = dict(headline=args.precision, arithmetic=PRECISIONS[args.precision][2], other_mode=other)
    x3 = (dict(ms=round(dt / args.steps * 1e3, 3), v=round(value, 2)) if args.precision == 'bf16x3'
          else (dict(ms=other.get('ms_per_step'), v=other.get('value')) if other else None))
    if x3:       # (the keys round 3's line carried for the reference-precision figure)
        prec.update(bf16x3_ms_per_step=x3['ms'], bf16x3_value=x3['v'], bf16x3_phases=args.mode)

    peaks = measured_peaks(dev) if rank == 0 else None
    roof = main_m['roof']
    if roof is not None:
        roof['measured_peaks'] = peaks
        if peaks and peaks.get('gemm_bf16_tflops'):
            roof['frac_of_measured_gemm'] = round(roof['achieved'] * PRECISIONS[args.precision][1] / peaks['gemm_bf16_tflops'], 4)
        if peaks and peaks.get('mfma_loop_tflops'):
            # issued MFMA rate of the dominant kernel class against what a register-resident MFMA loop delivers on THIS box at the clock the chip
            # holds under matrix load (the ceiling any kernel that also has to move operands stays under)
            roof['frac_of_measured_mfma_loop'] = round(roof['achieved'] * PRECISIONS[args.precision][1] / peaks['mfma_loop_tflops'], 4)
            for part in ('backbone_fpn', 'heads'):
                tot_ = (roof.get(part) or {}).get('total')
                if tot_:
                    tot_['frac_of_measured_mfma_loop'] = round(tot_['tflops'] * PRECISIONS[args.precision][1] / peaks['mfma_loop_tflops'], 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, cd, do_train, do_score)

    if rank == 0:
        line = dict(metric=cd['metric'], value=round(value, 2), unit='images/sec',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype=PRECISIONS[args.precision][0], data='synthetic',
                    config=dict(workload=f'{cd["name"]}, bs={B}/GPU: '
                                         + ' + '.join((['train iteration (main fwd/bwd/SGD + MEH fwd/bwd/SGD)'] if do_train else [])
                                                      + (['HUA scoring pass'] if do_score else [])),
                                value_definition=(f'SURVEY 8(d) AL-cycle rate N / (T_train + T_score): images per second through BOTH phases -- each step trains '
                                                  f'on {B} images/GPU and HUA-scores {B} images/GPU (other images of the same shape; pool scored by a frozen '
                                                  'copy with a calibrated, trained-like head); value = images/GPU * GPUs * steps / time') if phases == 2
                                else f'images per second through the {args.mode} phase',
                                value_both_phases=round(value * phases, 2),
                                global_batch=B * world, image_size=[H, W], num_classes=cd['classes'], backbone=f'ResNet-{cd["depth"]}',
                                parallelism=f'dp{world}', collective_ranks=comm['ranks'], collective_backend=comm['backend'],
                                phases=args.mode, launch='hip-graph replay' if use_graph else 'eager',
                                data_rotation=(f'{R} distinct resident train batches and {R} pool batches, cycled step by step through the graphs\' static input '
                                               'buffers (one 50 MB device copy + the ground-truth upload per phase and step, inside the timed region)' if R > 1
                                               else 'one resident train batch and one pool batch, placed in the static input buffers once'),
                                scores_read=('after the loop (the pool loop\'s deferred form: selection half of batch k on its own stream; one sync + one all-gather '
                                             'inside the timed region)' if main_m['deferred'] else 'after every step'),
                                ms_per_step_scores_read_every_step=main_m['serial_ms'],
                                arithmetic=PRECISIONS[args.precision][2],
                                column_sums='ordered partial sums (--deterministic)' if args.deterministic else 'fp32 atomics (default)'),
                    phase_rates=main_m['phase'], hua=hua, precision=prec, roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def pool_mode(args, cd, pool_model, dev, rank, world, B, H, W, cal_k, cal_frac, comm):
    """BASELINE configs[3]: HUA scoring of an unlabeled pool of --pool synthetic images, generated ON the device from
    Philox(seed=20, image id), through the product's own pool loop (apis/test.py single_gpu_uncertainty: contiguous shard per rank,
    HIP-graph replay per batch, ONE all-gather of the scores at the end).  A step = one batch of B images of this rank's shard."""
    from aod_meh_hua_amd.apis.test import single_gpu_uncertainty
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    from aod_meh_hua_amd.parallel import shard_range

    class Loader:       # what single_gpu_uncertainty reads from a DataLoader
        def __init__(self, ds):
            self.dataset, self.batch_size, self.collate_fn = ds, B, None
    kw = {k: v for k, v in SCORE_KW.items() if k not in ('return_loss', 'rescale', 'isEval', 'batchIdx')}
    pool_model.eval()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    # warm-up: a small pool of the same batch shape (captures the scoring graph inside the loop's GraphedScore on first use)
    with torch.no_grad():
        for _ in range(max(1, min(args.warmup, 2))):
            single_gpu_uncertainty(pool_model, Loader(DevicePhiloxPool(3 * B * world, (H, W), seed=21)), **kw)
        barrier()
        ds = DevicePhiloxPool(args.pool, (H, W), seed=20)
        lo, hi, per = shard_range(args.pool, rank, world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        unc = single_gpu_uncertainty(pool_model, Loader(ds), **kw)          # [N] on every rank (all-gathered)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    nb = -(-(hi - lo) // B)
    unc_h = unc.float().cpu()
    if rank == 0:
        sel = None
        try:
            from aod_meh_hua_amd.utils.active_datasets import update_X_L
            np.random.seed(20)
            n0 = max(args.pool // 20, 1)
            X_L, _ = update_X_L(unc_h.numpy().astype(np.float64), np.arange(args.pool), np.arange(n0), max(args.pool // 40, 1), zeroRate=0.15)
            sel = dict(X_L_next=len(X_L), first_selected=[int(v) for v in X_L[n0:n0 + 5]])
        except Exception:      # noqa: BLE001
            pass
        line = dict(metric='images/sec HUA pool scoring, RetinaNet-R50 VOC 512^2' if cd['depth'] == 50 else 'images/sec HUA pool scoring', value=round(args.pool / dt, 2),
                    unit='images/sec', n_gpus=world, steps=nb, warmup=args.warmup, ms_per_step=round(dt / max(nb, 1) * 1e3, 3), higher_is_better=True,
                    scaling='strong', vs_baseline=None, dtype=PRECISIONS[args.precision][0], data='synthetic',
                    config=dict(workload=f'{cd["name"]}: HUA unlabeled-pool scoring only, {args.pool} on-device Philox(seed=20, image id) images, '
                                         f'contiguous shard per rank, batches of {B}, one score all-gather',
                                pool=args.pool, global_batch=B * world, image_size=[H, W], num_classes=cd['classes'], parallelism=f'dp{world}',
                                collective_ranks=comm['ranks'], collective_backend=comm['backend'],
                                pool_partition=os.environ.get('AOD_POOL_SHARD', 'auto'), scores_sha16=__import__('hashlib').sha256(unc_h.numpy().tobytes()).hexdigest()[:16],
                                launch='hip-graph replay inside apis/test.py single_gpu_uncertainty'),
                    pool=dict(nonzero_scores=int((unc_h > 0).sum()), mean_score=round(float(unc_h.mean()), 5), gpu_ms=round(e0.elapsed_time(e1), 2),
                              wall_ms=round(dt * 1e3, 2), host_gap_ms=round(dt * 1e3 - e0.elapsed_time(e1), 2), selection=sel,
                              head_scale=round(cal_k, 3), fg_anchor_frac_at_calibration=round(cal_frac, 5)),
                    roofline=None, cpu_baseline=None)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(args, cd, do_train, do_score):
    """CPU oracle (port of the reference's CPU path, oracle/model.py) on a bounded sample: cells (threads, batch) =
    (2, 2) -- the reference's own torch.set_num_threads(2) and samples_per_gpu=2 (tools/train_RetinaNet.py:77, Config_RetinaNet.py:127) --,
    (16, 2), (all host cores capped at 64, 2) and (16, B); per cell the training iteration, the scoring forward and the HUA stage (pre-NMS +
    NMS + ComputeObjUnc on planted, trained-like head outputs so that the stage is not empty; timed at batch 2 and scaled by images for the
    batch-B cell) are timed separately.  `value` = the best cell's AL-cycle rate N / (T_train + T_score), computed like the GPU line's value.
    torch's CPU kernels stop scaling well below the 256 host threads of the GPU box, which is why the all-core cell is not the best one."""
    from oracle import model as om
    from tests import synth
    H, W, depth, nc = cd['H'], cd['W'], cd['depth'], cd['classes']
    budget = max(args.cpu_seconds, 5.0)
    ncpu = os.cpu_count() or 1
    cells = [(2, 2), (min(16, ncpu), 2), (min(ncpu, 64), 2), (min(16, ncpu), min(args.batch or cd['batch'], 16))]
    sd = om.seeded_state_dict(depth=depth, num_classes=nc)
    train_keys = [k for k, v in sd.items() if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.'))]
    for k in train_keys:
        sd[k].requires_grad_(True)
    meh = [k for k in train_keys if 'retina_L' in k or 'L_convs' in k]
    main = [k for k in train_keys if k not in meh]
    out_cells = []
    t_start = time.perf_counter()

    def timed(fn, share):
        """at least one call; repeat while the cell's share of the budget lasts"""
        t0, n = time.perf_counter(), 0
        while True:
            fn()
            n += 1
            d = time.perf_counter() - t0
            if d > share or n >= 5:
                return d / n
    for ci, (thr, Bc) in enumerate(cells):
        if time.perf_counter() - t_start > budget and out_cells:
            break
        torch.set_num_threads(thr)
        g = torch.Generator().manual_seed(20)
        img = torch.randn(Bc, 3, H, W, generator=g)
        gtb, gtl = synth.random_gts(Bc, H, W, seed=24, gmin=1, gmax=5, num_classes=nc)
        bufs, bufs_L = {}, {}

        def train_it():
            o = om.train_step(sd, img, gtb, gtl, depth=depth, num_classes=nc)
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

        def fwd_it():
            with torch.no_grad():
                feats = om.fpn(sd, om.backbone(sd, img, depth))
                om.head_forward(sd, feats)
                om.head_forward_L(sd, feats)
        heads = synth.planted_heads(2, H, W, C=nc, seed=22) if Bc == 2 else None

        def hua_it():
            with torch.no_grad():
                om.score_images(None, torch.zeros(2, 3, H, W), sampler='torch', heads=heads, num_classes=nc)
        share = budget / (len(cells) * 3)
        cell = dict(threads=thr, batch=Bc)
        if do_train:
            cell['train_s'] = round(timed(train_it, share), 3)
        if do_score:
            cell['score_forward_s'] = round(timed(fwd_it, share), 3)
            if Bc == 2:
                cell['hua_stage_s'] = round(timed(hua_it, share / 2), 3)
            else:       # same per-image work: scaled from the batch-2 cell with the same thread count
                ref = next((c for c in out_cells if c['threads'] == thr and c['batch'] == 2), out_cells[-1])
                cell['hua_stage_s'], cell['hua_stage_scaled_from_batch_2'] = round(ref['hua_stage_s'] * Bc / 2, 3), True
        tt = cell.get('train_s', 0.0) + cell.get('score_forward_s', 0.0) + cell.get('hua_stage_s', 0.0)
        cell['img_per_s'] = round(Bc / tt, 3)                   # AL-cycle rate, like the GPU line's value: Bc images through the timed phases
        out_cells.append(cell)
    best = max(out_cells, key=lambda c: c['img_per_s'])
    return dict(value=best['img_per_s'], unit='images/sec', cores=best['threads'], kind='port',
                sample=f'oracle/model.py (fp32 torch CPU ops) at {H}x{W}: per cell (threads, batch) one to five timed iterations of the training '
                       f'iteration, the scoring forward and the HUA stage (planted trained-like head outputs); value = best cell = AL-cycle rate, '
                       f'{best["batch"]} images trained and scored per ({best.get("train_s", 0)} + {best.get("score_forward_s", 0)} + {best.get("hua_stage_s", 0)}) s',
                host_cores=ncpu, cells=out_cells)


if __name__ == '__main__':
    main()
