"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: pool sharding + score all-gather (partition invariant,
padding trimmed) and gradient averaging through flat buckets."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aod_meh_hua_amd.parallel import GradSync, broadcast_model, gather_scores, shard_range
    lo, hi, per = shard_range(n_total)
    # each rank "scores" its block with a function of the GLOBAL image index only
    local = torch.tensor([float(i * i % 17) + 0.25 for i in range(lo, hi)])
    full = gather_scores(local, n_total)
    # the interleaved partition (batches of 3 striding over the ranks) scores the same images and gathers the same vector
    from aod_meh_hua_amd.parallel import gather_scores_indexed, shard_batches
    mine = [i for b in shard_batches(n_total, 3, interleaved=True) for i in b]
    full_i = gather_scores_indexed(torch.tensor([float(i * i % 17) + 0.25 for i in mine]), mine, n_total)
    assert torch.equal(full_i, full), (full_i, full)
    # ... and with the slot count the caller knows without communication (apis/test.py: ceil(ceil(N / bs) / world) * bs): ONE all-gather, no
    # all-reduce, no host read -- the form the pool loop uses
    per = -(-(-(-n_total // 3)) // world) * 3
    full_p = gather_scores_indexed(torch.tensor([float(i * i % 17) + 0.25 for i in mine]), mine, n_total, per=per)
    assert torch.equal(full_p, full), (full_p, full)
    assert [i for b in shard_batches(n_total, 3) for i in b] == list(range(lo, hi))       # contiguous form = the shard_range block
    # gradient averaging: rank r holds grads r+1 on two tensors spanning two buckets
    p1, p2 = torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))
    p1.grad, p2.grad = torch.full((5,), float(rank + 1)), torch.full((3, 2), float(10 * (rank + 1)))
    gs = GradSync(bucket_mb=1)
    gs.bucket_elems = 6
    gs.all_reduce_grads([p1, p2])
    m = torch.nn.Linear(2, 2)
    with torch.no_grad():
        m.weight.fill_(float(rank))
    v0 = m.weight._version
    broadcast_model(m)
    assert m.weight._version > v0                       # packed-weight caches keyed on the version see the broadcast values
    # segmented start: buckets never straddle a backward segment; each segment is reduced on its own, in completion order
    qs = [torch.nn.Parameter(torch.zeros(n)) for n in (4, 70, 3, 5)]
    gs2 = GradSync(bucket_mb=1)
    ent = gs2.attach(qs, segments=[[qs[3]], [qs[1], qs[2]], [qs[0]]])      # (heads), (deep stage), (shallow stage)
    assert gs2.num_segments(qs) == 3 and sorted(b['seg'] for b in ent['buckets']) == [0, 1, 2]
    for k, idx in ((0, [3]), (1, [1, 2]), (2, [0])):
        for i in idx:
            qs[i].grad = torch.full_like(qs[i], float((rank + 1) * (i + 1)))
        gs2.start(qs, segment=k).wait()
        for i in idx:
            assert qs[i].grad.data_ptr() == qs[i]._aod_grad_view.data_ptr() and torch.equal(qs[i].grad, torch.full_like(qs[i], 1.5 * (i + 1)))
        for i in range(4):                                # later segments are untouched so far
            if i not in idx and qs[i].grad is None:
                assert float(qs[i]._aod_grad_view.abs().sum()) == 0.0
    q.put((rank, lo, hi, full.tolist(), p1.grad.tolist(), p2.grad.flatten().tolist(), m.weight.flatten().tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [7, 8, 1])
def test_shard_gather_and_grad_sync_world2(n_total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [float(i * i % 17) + 0.25 for i in range(n_total)]
    covered = []
    for rank, lo, hi, full, g1, g2, w in res:
        assert full == expect                               # identical on every rank, padding trimmed
        covered += list(range(lo, hi))
        assert g1 == [1.5] * 5 and g2 == [15.0] * 6         # mean of per-rank gradients
        assert w == [0.0] * 4                               # rank-0 weights everywhere
    assert covered == list(range(n_total))                  # blocks tile the pool exactly once


def test_single_process_paths_are_noops():
    sys.path.insert(0, ROOT)
    from aod_meh_hua_amd.parallel import GradSync, gather_scores, get_dist_info, shard_range
    assert get_dist_info() == (0, 1) and shard_range(10) == (0, 10, 10)
    x = torch.arange(5.0)
    assert torch.equal(gather_scores(x, 5), x)
    from aod_meh_hua_amd.parallel import gather_scores_indexed, shard_batches
    assert shard_batches(7, 3, 0, 2, interleaved=True) == [[0, 1, 2], [6]] and shard_batches(7, 3, 1, 2, interleaved=True) == [[3, 4, 5]]
    assert torch.equal(gather_scores_indexed(torch.tensor([1., 2., 3.]), [4, 0, 2], 5), torch.tensor([2., 0., 3., 0., 1.]))
    GradSync().all_reduce_grads([torch.nn.Parameter(torch.zeros(2))])
