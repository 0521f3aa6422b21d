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


class GradSync:
    def __init__(self, bucket_mb=64):
        self.bucket_elems = bucket_mb * (1 << 20) // 4

    def all_reduce_grads(self, params):
        if not is_dist():
            return
        world = dist.get_world_size()
        grads = [p.grad for p in params if p.grad is not None]
        i = 0
        while i < len(grads):
            j, n = i, 0
            while j < len(grads) and (n == 0 or n + grads[j].numel() <= self.bucket_elems):
                n += grads[j].numel()
                j += 1
            flat = torch.cat([g.reshape(-1).float() for g in grads[i:j]])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            o = 0
            for g in grads[i:j]:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()
            i = j


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
