"""One-process-per-GPU data parallelism over RCCL (torch.distributed backend 'nccl' == RCCL on ROCm; 'gloo'
in the CPU tests).  The reference has no working multi-GPU training path (it wraps MMDataParallel,
apis/train_Lambda.py:53, and scores the pool with dist=False, tools/train_RetinaNet.py:224-225); SURVEY 8(e)
defines what is built here:

  * GradSync      -- mean of per-rank gradients: parameters' grads are packed into a few large flat fp32
                     buckets (sized for xGMI's per-link-bound rings: few, large collectives), all-reduced,
                     and unpacked; called once after each of the two backward passes of run_iter.
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
    """In-flight bucketed all-reduce: wait() finishes the collectives and scatters the averaged buckets back into the gradients."""

    def __init__(self, buckets, world):
        self.buckets, self.world = buckets, world

    def wait(self):
        for grads, flat, work in self.buckets:
            if work is not None:
                work.wait()
            flat.div_(self.world)
            parts = [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)]
            if hasattr(torch, '_foreach_copy_'):
                torch._foreach_copy_(grads, parts)          # one batched launch instead of one copy per parameter
            else:
                for g, v in zip(grads, parts):
                    g.copy_(v)
        self.buckets = []


class GradSync:
    def __init__(self, bucket_mb=64):
        self.bucket_elems = bucket_mb * (1 << 20) // 4

    def start(self, params):
        """Launch the bucketed all-reduce of the parameters' gradients WITHOUT waiting (RCCL runs it on its own stream; xGMI rings are
        per-link bound, so a few large buckets).  Returns a handle whose wait() must be called before the gradients are read.  The
        runner overlaps the main network's all-reduce with the whole MEH forward/backward (disjoint parameters)."""
        if not is_dist():
            return _PendingSync([], 1)
        world = dist.get_world_size()
        grads = [p.grad for p in params if p.grad is not None]
        buckets, i = [], 0
        while i < len(grads):
            j, n = i, 0
            while j < len(grads) and (n == 0 or n + grads[j].numel() <= self.bucket_elems):
                n += grads[j].numel()
                j += 1
            flat = torch.cat([g.reshape(-1).float() for g in grads[i:j]])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            buckets.append((grads[i:j], flat, work))
            i = j
        return _PendingSync(buckets, world)

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
