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
from ..parallel import gather_scores, get_dist_info, shard_range


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


def calculate_uncertainty(cfg, *args, **kwargs):
    """test.py:65-70."""
    return getattr(Uncertainty_fns, cfg.uncertainty_pool)(cfg, *args, **kwargs)


def _unwrap(x):
    return x.data if isinstance(x, DataContainer) else x


def single_gpu_uncertainty(model, data_loader, **kwargs):
    """test.py:90-135, sharded.  Returns a [N] fp32 tensor (N = len(dataset)) identical on every rank."""
    model.eval()
    dataset = data_loader.dataset
    N = len(dataset)
    rank, world = get_dist_info()
    lo, hi, per = shard_range(N, rank, world)
    bs = data_loader.batch_size or 1
    collate = data_loader.collate_fn
    prog_bar = ProgressBar(hi - lo)
    chunks, conf_chunks = [], []
    kwargs.setdefault('scaleUnc', False)
    # HIP-graph replay of the scoring batch (graphs.GraphedScore) while batches keep one shape; side-effect options stay eager
    import os
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
    all_ids = torch.arange(lo, max(hi, lo), dtype=torch.int64).to(dev) if device_side else None
    for s in range(lo, hi, bs):
        idxs = list(range(s, min(s + bs, hi)))
        if device_side:
            image_ids = all_ids[s - lo:s - lo + len(idxs)]
            data = dataset.device_batch(idxs, dev, image_ids=image_ids)
        else:
            data = collate([dataset[i] for i in idxs])
            data = {k: _unwrap(v) for k, v in data.items() if k in ('img', 'img_metas')}
            image_ids = torch.tensor(idxs, dtype=torch.int64).to(dev, non_blocking=True)
        out = None
        if gscore is not None and isinstance(data['img'], (list, tuple)) and len(data['img']) == 1:
            out = gscore.maybe(data['img'][0], data['img_metas'][0], image_ids)
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
    local = torch.cat(chunks) if chunks else torch.zeros(0, device=dev)
    if kwargs.get('saveMaxConf'):
        conf = torch.cat(conf_chunks) if conf_chunks else torch.zeros(0, device=dev)
        return gather_scores(local, N), gather_scores(conf, N)
    return gather_scores(local, N)


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
