"""One-process-per-GPU data parallelism over RCCL (torch.distributed backend 'nccl' == RCCL on ROCm; 'gloo'
in the CPU tests).  The reference has no working multi-GPU training path (it wraps MMDataParallel,
apis/train_Lambda.py:53, and scores the pool with dist=False, tools/train_RetinaNet.py:224-225); SURVEY 8(e)
defines what is built here:

  * GradSync      -- mean of per-rank gradients: each optimizer's gradients live in ONE persistent flat fp32
                     buffer (parameters' .grad are slices of it), all-reduced in place in a few large buckets
                     (sized for xGMI's per-link-bound rings); called once after each of the two backward passes.
  * shard_range   -- contiguous block of the unlabeled pool for this rank, [r*ceil(N/W), (r+1)*ceil(N/W)).
  * gather_scores -- all-gather of the per-rank fp32 score blocks (+ trim of the padding).
  * broadcast_model -- rank-0 weights to everybody after each cycle's re-init (tools/train_RetinaNet.py:156-157).
"""
import math

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _PendingSync:
    """In-flight bucketed all-reduce of (a segment of) one flat gradient buffer: wait() finishes the collectives, turns the sums into means
    and makes every covered parameter's .grad the averaged slice of the flat buffer (a re-pointing, not a copy)."""

    def __init__(self, ent, works, world, avg_done, idx=None, ranges=None):
        self.ent, self.works, self.world, self.avg_done, self.idx, self.ranges = ent, works, world, avg_done, idx, ranges

    def wait(self):
        if self.ent is None:
            return
        for w in self.works:
            if w is not None:
                w.wait()
        if not self.avg_done and self.world > 1:
            for lo, hi in self.ranges:
                self.ent['flat'][lo:hi].mul_(1.0 / self.world)
        idx = self.idx if self.idx is not None else range(len(self.ent['params']))
        for i in idx:
            p, v = self.ent['params'][i], self.ent['views'][i]
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        self.ent = None


class _PendingList:
    def __init__(self, items):
        self.items = items

    def wait(self):
        for it in self.items:
            it.wait()
        self.items = []


class GradSync:
    """Mean of per-rank gradients over RCCL.  Each parameter list (one per optimizer) owns ONE persistent flat fp32 buffer; every
    parameter's gradient is a slice of it (`p._aod_grad_view`).  The conv backward writes weight gradients straight into the slice
    (functional.ConvFn), so for 99.8 % of the bytes .grad IS the slice and the all-reduce runs in place: no torch.cat, no copy back
    (round 1 moved ~3 x 146 MB per iteration for that).  The few small vectors autograd hands over in their own tensors (BN gamma / beta,
    biases: ~60 K floats) are copied into their slices by one batched launch.  The buffer is reduced in a few large buckets -- xGMI rings
    are per-link bound -- that never straddle a backward SEGMENT (attach(..., segments=...): head + neck, then the backbone stages from the
    deepest to the shallowest -- the order functional.backward_segments() finishes them in): start(..., segment=k) launches the buckets of
    segment k as soon as its gradients are final, while the next segment's backward runs (the reference's own pattern for
    gradient buckets: mmdet/core/utils/dist_utils.py:10-51)."""

    ALIGN = 64          # elements: every slice starts on a 256-B boundary (vector stores of the unpack kernel)

    def __init__(self, bucket_mb=64):
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._ents = {}
        # the all-reduces of this object run on RCCL's stream BESIDE the backward pass: keep the backward launches off the persistent
        # one-workgroup-per-CU conv kernel (csrc/conv_x3p.hip, aod_conv_x3p_wants: a grid that assumes every CU free doubles its time when RCCL's
        # channel workgroups hold a few); forward and scoring launches are not concerned.  An explicit AOD_X3P_DGRAD in the environment wins.
        if is_dist() and dist.get_world_size() > 1:
            import os
            os.environ.setdefault('AOD_X3P_DGRAD', '0')

    def attach(self, params, segments=None):
        """segments: list of parameter lists (completion order of the backward segments); None keeps what an earlier attach() fixed."""
        params = [p for p in params if p.requires_grad]
        key = tuple(id(p) for p in params)
        seg_of = None
        if segments is not None and len(segments) > 1:
            where = {id(q): k for k, g in enumerate(segments) for q in g}
            seg_of = tuple(where[id(p)] for p in params)
        ent = self._ents.get(key)
        if ent is not None and all(r() is p for r, p in zip(ent['refs'], params)) and (segments is None or ent['seg_of'] == seg_of):
            return ent
        import weakref
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        flat = ent['flat'] if ent is not None and ent['flat'].numel() == max(n, 1) and ent['flat'].device == params[0].device else \
            torch.zeros(max(n, 1), dtype=torch.float32, device=params[0].device)
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
        for p, v in zip(params, views):
            p._aod_grad_view = v
        # buckets = contiguous element ranges, cut at parameter boundaries -- and at segment boundaries --, walking the parameters backwards
        sg = seg_of if seg_of is not None else (0,) * len(params)
        buckets, hi, i = [], n, len(params) - 1
        while i >= 0:
            lo = hi
            first = i
            while i >= 0 and sg[i] == sg[first] and (lo == hi or hi - offs[i] <= self.bucket_elems):
                lo = offs[i]
                i -= 1
            buckets.append(dict(lo=lo, hi=hi, first_param=i + 1, last_param=first, seg=sg[first]))
            hi = lo
        ent = dict(flat=flat, views=views, params=params, refs=[weakref.ref(p) for p in params], offs=offs, buckets=buckets, seg_of=seg_of,
                   nseg=(max(sg) + 1) if params else 1)
        self._ents[key] = ent
        return ent

    def _gather_small(self, ent, sources=None, idx=None):
        """Copy the gradients that do not already live in their slices into them (one batched launch); zero the slices of parameters
        without a gradient this iteration."""
        srcs, dsts = [], []
        for i in (idx if idx is not None else range(len(ent['params']))):
            p, v = ent['params'][i], ent['views'][i]
            g = sources[i] if sources is not None else p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                srcs.append(g.reshape(v.shape) if g.dtype == torch.float32 else g.reshape(v.shape).float())
                dsts.append(v)
            p.__dict__.pop('_aod_view_busy', None)
        if srcs:
            if hasattr(torch, '_foreach_copy_'):
                torch._foreach_copy_(dsts, srcs)
            else:
                for d, s_ in zip(dsts, srcs):
                    d.copy_(s_)

    def release(self, params):
        """Hand the parameters' flat-buffer slices back to the next backward pass WITHOUT communicating (graph warm-up iterations):
        the bookkeeping half of start()."""
        for p in params:
            p.__dict__.pop('_aod_view_busy', None)

    def num_segments(self, params):
        return self.attach(params)['nseg']

    def start(self, params, sources=None, segment=None):
        """Launch the bucketed all-reduce of the parameters' gradients WITHOUT waiting (RCCL runs it on its own stream).  Returns a
        handle whose wait() must be called before the gradients are read.  `sources`: gradient tensors to read instead of p.grad (the
        static tensors a captured HIP graph writes, graphs.GraphedTrainStep; indexed like the parameter list).  `segment`: only the buckets
        of that backward segment (its gradients are final; later segments are still being computed).  The runner also overlaps the main
        network's last buckets with the whole MEH forward/backward (disjoint parameters)."""
        if not is_dist():
            return _PendingSync(None, [], 1, True)
        ent = self.attach(params)
        idx = None
        if segment is not None and ent['seg_of'] is not None:
            idx = [i for i, k in enumerate(ent['seg_of']) if k == segment]
        self._gather_small(ent, sources, idx)
        world = dist.get_world_size()
        avg = dist.get_backend() == 'nccl'
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        bks = [b for b in ent['buckets'] if b['hi'] > b['lo'] and (idx is None or b['seg'] == segment)]
        works = [dist.all_reduce(ent['flat'][b['lo']:b['hi']], op=op, async_op=True) for b in bks]
        return _PendingSync(ent, works, world, avg, idx, [(b['lo'], b['hi']) for b in bks])

    def all_reduce_grads(self, params):
        self.start(params).wait()


def backward_and_sync(gsync, params, loss, cuts):
    """loss.backward() in segments (functional.backward_segments) with each segment's bucket all-reduce launched as soon as its gradients
    are final -- the later segments' backward kernels then run beside RCCL.  Returns a handle: wait() before the optimizer step."""
    from . import functional as AF
    pend = []
    if not is_dist() or gsync.num_segments(params) != len(cuts) + 1:
        AF.backward_segments(loss, cuts)
        pend.append(gsync.start(params))
    else:
        AF.backward_segments(loss, cuts, after=lambda k: pend.append(gsync.start(params, segment=k)))
    return _PendingList(pend)


def shard_range(n_total, rank=None, world=None):
    """Contiguous block per rank; the last blocks may be short or empty."""
    if rank is None:
        rank, world = get_dist_info()
    per = math.ceil(n_total / world) if n_total else 0
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total), per


def shard_batches(n_total, bs, rank=None, world=None, interleaved=False):
    """This rank's batches of the pool as lists of GLOBAL image indices.  Contiguous (SURVEY 8e default): the shard_range block cut into
    batches.  Interleaved: global batch k (images [k*bs, (k+1)*bs)) goes to rank k mod W -- the per-image work of the HUA stage varies with
    the number of (candidate, object) pairs, and pools are often ordered (by video, by scene): striding the batches evens that out
    ('interleaved rather than contiguous assignment if imbalance > 5 %').  The image ids key the Philox streams, so either
    partition gives bit-identical scores."""
    if rank is None:
        rank, world = get_dist_info()
    if not interleaved:
        lo, hi, _ = shard_range(n_total, rank, world)
        return [list(range(s, min(s + bs, hi))) for s in range(lo, hi, bs)]
    nb = math.ceil(n_total / bs) if n_total else 0
    return [list(range(k * bs, min((k + 1) * bs, n_total))) for k in range(rank, nb, world)]


def gather_scores_indexed(local_scores, local_idx, n_total, per=None):
    """Scores of an arbitrary index set per rank -> the full [N] vector on every rank with ONE all-gather: each rank sends a [per, 2] fp32
    block whose column 0 holds its scores and whose column 1 holds the global indices as int32 bit patterns (pad slots: -1).
    per: slots per rank, the same number on every rank (the interleaved partition's bound ceil(ceil(N / bs) / world) * bs is known to the
    caller without communication); None -> agreed on by one extra all-reduce(MAX) and a host read of the result."""
    rank, world = get_dist_info()
    dev = local_scores.device
    idx = torch.as_tensor(local_idx, dtype=torch.int64, device=dev)
    full = torch.zeros(n_total, dtype=local_scores.dtype, device=dev)
    if world == 1:
        full[idx] = local_scores
        return full
    assert n_total < (1 << 31)
    if per is None:
        cnt = torch.tensor([idx.numel()], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        per = int(cnt)
    assert idx.numel() <= per, (idx.numel(), per)
    buf = torch.zeros(per, 2, dtype=torch.float32, device=dev)
    ib = buf.view(torch.int32)
    ib[:, 1] = -1
    buf[:idx.numel(), 0] = local_scores.float()
    ib[:idx.numel(), 1] = idx.to(torch.int32)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    allb = torch.cat(out)
    ix = allb.view(torch.int32)[:, 1].long()
    ok = ix >= 0
    full[ix[ok]] = allb[ok, 0].to(full.dtype)
    return full


def gather_scores(local_scores, n_total):
    """local_scores: 1-D fp32 tensor of this rank's block (len <= ceil(N/W)).  Returns the full [N] tensor on
    every rank."""
    rank, world = get_dist_info()
    if world == 1:
        return local_scores[:n_total]
    per = math.ceil(n_total / world)
    buf = local_scores.new_zeros(per)
    buf[:local_scores.numel()] = local_scores
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat(out)[:n_total]


def broadcast_model(model, src=0):
    """rank `src`'s parameters and buffers to every rank: ONE broadcast per dtype of a flat copy (a ResNet-50 detector has ~540 tensors;
    one collective each is ~540 launches per active-learning cycle), scattered back by a batched copy"""
    if not is_dist():
        return
    by = {}
    for t in list(model.parameters()) + list(model.buffers()):
        by.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():                 # (in-place on the tensors themselves, not .data: the version counters move, so packed-weight
        for (dt, dev), ts in by.items():  # caches keyed on them -- functional.ParamPrep -- see the new values)
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src)
            outs = [o.view_as(t) for o, t in zip(flat.split([t.numel() for t in ts]), ts)]
            if hasattr(torch, '_foreach_copy_') and dt.is_floating_point:
                torch._foreach_copy_(ts, outs)
            else:
                for t, o in zip(ts, outs):
                    t.copy_(o)
