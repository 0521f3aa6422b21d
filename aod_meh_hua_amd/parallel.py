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
    """In-flight bucketed all-reduce of one flat gradient buffer: wait() finishes the collectives, turns the sums into means and makes
    every parameter's .grad the averaged slice of the flat buffer (a re-pointing, not a copy)."""

    def __init__(self, ent, works, world, avg_done):
        self.ent, self.works, self.world, self.avg_done = ent, works, world, avg_done

    def wait(self):
        if self.ent is None:
            return
        for w in self.works:
            if w is not None:
                w.wait()
        if not self.avg_done and self.world > 1:
            self.ent['flat'].mul_(1.0 / self.world)
        for p, v in zip(self.ent['params'], self.ent['views']):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        self.ent = None


class GradSync:
    """Mean of per-rank gradients over RCCL.  Each parameter list (one per optimizer) owns ONE persistent flat fp32 buffer; every
    parameter's gradient is a slice of it (`p._aod_grad_view`).  The conv backward writes weight gradients straight into the slice
    (functional.ConvFn), so for 99.8 % of the bytes .grad IS the slice and the all-reduce runs in place: no torch.cat, no copy back
    (round 1 moved ~3 x 146 MB per iteration for that).  The few small vectors autograd hands over in their own tensors (BN gamma / beta,
    biases: ~60 K floats) are copied into their slices by one batched launch.  The buffer is reduced in a few large buckets -- xGMI rings
    are per-link bound -- issued in REVERSE parameter order: the order in which a backward pass finishes them (`bucket_ready` hooks use
    that to start a bucket's all-reduce while the rest of the backward is still running)."""

    ALIGN = 64          # elements: every slice starts on a 256-B boundary (vector stores of the unpack kernel)

    def __init__(self, bucket_mb=64):
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._ents = {}

    def attach(self, params):
        params = [p for p in params if p.requires_grad]
        key = tuple(id(p) for p in params)
        ent = self._ents.get(key)
        if ent is not None and all(r() is p for r, p in zip(ent['refs'], params)):
            return ent
        import weakref
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        flat = torch.zeros(max(n, 1), dtype=torch.float32, device=params[0].device)
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
        for p, v in zip(params, views):
            p._aod_grad_view = v
        # buckets = contiguous element ranges, cut at parameter boundaries, walking the parameters backwards
        buckets, hi, i = [], n, len(params) - 1
        while i >= 0:
            lo = hi
            first = i
            while i >= 0 and (lo == hi or hi - offs[i] <= self.bucket_elems):
                lo = offs[i]
                i -= 1
            buckets.append(dict(lo=lo, hi=hi, first_param=i + 1, last_param=first))
            hi = lo
        ent = dict(flat=flat, views=views, params=params, refs=[weakref.ref(p) for p in params], offs=offs, buckets=buckets)
        self._ents[key] = ent
        return ent

    def _gather_small(self, ent, sources=None):
        """Copy the gradients that do not already live in their slices into them (one batched launch); zero the slices of parameters
        without a gradient this iteration."""
        srcs, dsts = [], []
        for i, (p, v) in enumerate(zip(ent['params'], ent['views'])):
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

    def start(self, params, sources=None):
        """Launch the bucketed all-reduce of the parameters' gradients WITHOUT waiting (RCCL runs it on its own stream).  Returns a
        handle whose wait() must be called before the gradients are read.  `sources`: gradient tensors to read instead of p.grad (the
        static tensors a captured HIP graph writes, graphs.GraphedTrainStep).  The runner overlaps the main network's all-reduce with the
        whole MEH forward/backward (disjoint parameters)."""
        if not is_dist():
            return _PendingSync(None, [], 1, True)
        ent = self.attach(params)
        self._gather_small(ent, sources)
        world = dist.get_world_size()
        avg = dist.get_backend() == 'nccl'
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works = [dist.all_reduce(ent['flat'][b['lo']:b['hi']], op=op, async_op=True) for b in ent['buckets'] if b['hi'] > b['lo']]
        return _PendingSync(ent, works, world, avg)

    def all_reduce_grads(self, params):
        self.start(params).wait()


def shard_range(n_total, rank=None, world=None):
    """Contiguous block per rank; the last blocks may be short or empty."""
    if rank is None:
        rank, world = get_dist_info()
    per = math.ceil(n_total / world) if n_total else 0
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total), per


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
    if not is_dist():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)
