"""calculate_uncertainty / single_gpu_uncertainty / single_gpu_test with the reference's names and kwargs
(mmdet/apis/test.py:19-195).

Re-design for 8 x MI355X: the pool is SHARDED -- each rank scores the contiguous block
[r*ceil(N/W), (r+1)*ceil(N/W)) of the loader's dataset and the per-image fp32 scores are all-gathered over RCCL
(the reference's loader is dist=False, so every rank would score the whole pool).  Scores stay on the device
inside the loop; the host syncs once per pool.  The Philox stream is keyed by the global image index, so the
result is bit-identical for any world size."""
import numpy as np
import torch

from ..mmcv_lite import DataContainer, ProgressBar
from ..parallel import gather_scores, gather_scores_indexed, get_dist_info, shard_batches, shard_range


class Uncertainty_fns:
    @staticmethod
    def Random(cfg, *args, **kwargs):
        return torch.randperm(len(args[1].dataset)).numpy()

    @staticmethod
    @torch.no_grad()
    def Entropy_NMS(cfg, *args, **kwargs):
        model, dataloader = args
        model.eval()
        unc = single_gpu_uncertainty(model, dataloader, isUnc=cfg.uncertainty_type, uPool=cfg.uncertainty_pool,
                                     uPool2=cfg.uncertainty_pool2, **kwargs)
        return unc.cpu() if torch.is_tensor(unc) else [u.cpu() if torch.is_tensor(u) else u for u in unc]

    Entropy_ALL = Entropy_NMS          # test.py:52-63: same loop; the head switches on cfg.uncertainty_pool (ComputeScaleUnc path)

    @staticmethod
    def Entropy_NoNMS(cfg, *args, **kwargs):
        # the reference's own Entropy_NoNMS path calls ComputeScaleUnc with L_scores=None (Lambda_L2.py:404-405,364) and raises a
        # TypeError on the first batch: there is no behaviour to reproduce
        raise NotImplementedError('uncertainty_pool=Entropy_NoNMS is not runnable in the reference either (L_scores is None there)')


# captured scoring graphs, per model (weak keys; a GraphedScore holds its model weakly too, so a dead model frees its graph memory pool by
# reference counting -- and nothing unpicklable hangs in the module's __dict__: copy.deepcopy(model) keeps working after a pool was scored)
import weakref as _weakref
_GSCORE = _weakref.WeakKeyDictionary()


_SHARD_STATE = dict(interleaved=False, imbalance=None, over=0)       # pool partition across ranks (single_gpu_uncertainty)


def calculate_uncertainty(cfg, *args, **kwargs):
    """test.py:65-70."""
    return getattr(Uncertainty_fns, cfg.uncertainty_pool)(cfg, *args, **kwargs)


def _unwrap(x):
    return x.data if isinstance(x, DataContainer) else x


def _shard_batches(dataset, batches, collate, workers):
    """(idxs, collated batch) over this rank's batches (lists of global image indices) of the pool.  workers > 0: worker processes decode / resize / normalise the
    images ahead of the GPU (the reference builds its pool loader with cfg.data.workers_per_gpu workers, tools/train_RetinaNet.py:224-225,
    mmdet/datasets/builder.py:76-139) into PINNED host memory, `prefetch` batches deep, so that the H2D copy of batch i + 1 and the host work
    of batches i + 2.. overlap the scoring graph of batch i; at ~4 000 images/s of GPU rate a single-process PIL decode + resize would bound
    the pool loop by more than 10x.  workers == 0: the synchronous loop (same batches, same order)."""
    if workers <= 0 or not batches:
        for idxs in batches:
            yield idxs, collate([dataset[i] for i in idxs])
        return
    from torch.utils.data import DataLoader
    dl = DataLoader(dataset, batch_sampler=batches, num_workers=workers, collate_fn=collate, pin_memory=False, prefetch_factor=4)
    for idxs, batch in zip(batches, dl):
        yield idxs, batch


def _pin(x):
    """pinned copy of a host tensor (asynchronous H2D source); DataContainers / lists are walked"""
    if torch.is_tensor(x):
        return x.pin_memory() if x.device.type == 'cpu' and not x.is_pinned() else x
    if isinstance(x, (list, tuple)):
        return type(x)(_pin(v) for v in x)
    return x


def single_gpu_uncertainty(model, data_loader, **kwargs):
    """test.py:90-135, sharded.  Returns a [N] fp32 tensor (N = len(dataset)) identical on every rank."""
    model.eval()
    dataset = data_loader.dataset
    N = len(dataset)
    rank, world = get_dist_info()
    bs = data_loader.batch_size or 1
    # partition of the pool over the ranks (parallel.shard_batches): AOD_POOL_SHARD = contiguous | interleaved | auto (default: contiguous
    # until a pass measured more than 5 % imbalance between the ranks' loop times -- then every later pass of this process strides the batches)
    import os
    shard_mode = os.environ.get('AOD_POOL_SHARD', 'auto')
    interleaved = shard_mode == 'interleaved' or (shard_mode == 'auto' and _SHARD_STATE['interleaved'])
    my_batches = shard_batches(N, bs, rank, world, interleaved)
    collate = data_loader.collate_fn
    prog_bar = ProgressBar(sum(len(b) for b in my_batches))
    chunks, conf_chunks = [], []
    kwargs.setdefault('scaleUnc', False)
    # HIP-graph replay of the scoring batch (graphs.GraphedScore) while batches keep one shape; side-effect options stay eager
    plain = not any(kwargs.get(k) for k in ('showNMS', 'saveUnc', 'saveMaxConf', 'scaleUnc', 'draw'))
    others = []
    gscore = None
    if plain and os.environ.get('AOD_HIP_GRAPH', '1') != '0' and next(model.parameters()).is_cuda:
        from ..graphs import GraphedScore
        # one captured graph per (model, options): a pool is scored once per AL cycle with a freshly built model, but callers that score
        # several pools with one model (bench, tests) must not pay the capture again
        # (the key holds hashable primitives only; any other option value -> no caching, no graph: str() of a tensor / object could collide)
        if all(isinstance(v, (bool, int, float, str, type(None))) for v in kwargs.values()):
            cache = _GSCORE.setdefault(model, {})
            key = tuple(sorted((k, type(v).__name__, v) for k, v in kwargs.items()))
            gscore = cache.get(key)
            if gscore is None:
                gscore = cache[key] = GraphedScore(model, rescale=True, isEval=False, batchIdx=0, **kwargs)
    dev = next(model.parameters()).device
    device_side = hasattr(dataset, 'device_batch')       # images produced on the device (datasets.DevicePhiloxPool): no host collate / H2D
    my_idx = [i for b in my_batches for i in b]
    all_ids = torch.tensor(my_idx, dtype=torch.int64).to(dev)             # GLOBAL image ids: they key the Philox streams of the HUA sampler
    workers = int(os.environ.get('AOD_POOL_WORKERS', getattr(data_loader, 'num_workers', 0) or 0))
    batches = ((idxs, None) for idxs in my_batches) if device_side else _shard_batches(dataset, my_batches, collate, workers)
    timed = world > 1 and dev.type == 'cuda' and torch.distributed.is_available() and torch.distributed.is_initialized()
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    pos = 0
    for idxs, data in batches:
        s = idxs[0]
        image_ids = all_ids[pos:pos + len(idxs)]
        pos += len(idxs)
        if device_side:
            out = gscore.static_image((len(idxs), 3) + tuple(dataset.size)) if gscore is not None and hasattr(dataset, 'size') else None
            data = dataset.device_batch(idxs, dev, image_ids=image_ids, out=out)
        else:
            data = {k: _unwrap(v) for k, v in data.items() if k in ('img', 'img_metas')}
            if dev.type == 'cuda':
                data['img'] = _pin(data['img'])
        out = None
        if gscore is not None and isinstance(data['img'], (list, tuple)) and len(data['img']) == 1:
            out = gscore.maybe(data['img'][0], data['img_metas'][0], image_ids, defer=True)       # (scores are read after gscore.sync() below)
        if out is not None:
            result, unc = out
        else:
            with torch.no_grad():
                result, unc, *others = model(return_loss=False, rescale=True, isEval=False, batchIdx=s // bs, image_ids=image_ids, **data, **kwargs)
        chunks.append(torch.as_tensor(unc, dtype=torch.float32, device=dev).reshape(-1))
        if kwargs.get('saveMaxConf'):          # test.py:130,134
            conf_chunks.append(torch.as_tensor(others[0], dtype=torch.float32, device=dev).reshape(-1))
        prog_bar.update(len(idxs))
    dev = next(model.parameters()).device
    if gscore is not None:
        gscore.sync()                       # deferred selection halves (second stream) -> this stream
    local = torch.cat(chunks) if chunks else torch.zeros(0, device=dev)
    if timed:           # load balance of this pass: every rank learns every rank's loop time and takes the same decision for the next pass
        import torch.distributed as dist
        ev1.record()
        ev1.synchronize()
        t = torch.tensor([ev0.elapsed_time(ev1)], device=dev)
        ts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        ts = torch.cat(ts)
        _SHARD_STATE['imbalance'] = float((ts.max() - ts.min()) / ts.max().clamp_min(1e-9))
        # (two CONSECUTIVE passes over the threshold: a pass that captured its scoring graph or started its loader workers measures that, not
        # the pool -- ADVICE r4)
        _SHARD_STATE['over'] = _SHARD_STATE['over'] + 1 if _SHARD_STATE['imbalance'] > 0.05 else 0
        if _SHARD_STATE['over'] >= 2:
            _SHARD_STATE['interleaved'] = True
    per = -(-(-(-N // bs)) // world) * bs       # slots per rank of the interleaved partition: ceil(ceil(N / bs) / world) batches of bs
    gather = (lambda v: gather_scores_indexed(v, my_idx, N, per=per)) if interleaved else (lambda v: gather_scores(v, N))
    if kwargs.get('saveMaxConf'):
        conf = torch.cat(conf_chunks) if conf_chunks else torch.zeros(0, device=dev)
        return gather(local), gather(conf)
    return gather(local)


@torch.no_grad()
def single_gpu_test(model, data_loader, show=False, out_dir=None, show_score_thr=0.3, **kwargs):
    """test.py:138-195 (detection results for evaluation; isEval=True)."""
    model.eval()
    results = []
    for data in data_loader:
        data = {k: _unwrap(v) for k, v in data.items() if k in ('img', 'img_metas')}
        kw = dict(kwargs)
        kw.setdefault('isUnc', False)              # EvalHook forwards the whole `evaluation` dict (interval popped): isUnc, metric, ...
        results.extend(model(return_loss=False, rescale=True, isEval=True, **data, **kw))
    return results
