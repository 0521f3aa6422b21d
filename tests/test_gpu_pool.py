"""GPU: BASELINE configs[3] -- HUA scoring of an on-device Philox pool through the product's own pool loop
(apis/test.py single_gpu_uncertainty <- mmdet/apis/test.py:90-135; caller tools/train_RetinaNet.py:221-246).

  * datasets.DevicePhiloxPool images == the numpy restatement of the Philox stream (oracle/pool.py): a pool image is a pure function of
    (seed, image id);
  * the scores of a 200-image 512x512 pool are BIT-equal for batch sizes 16 and 5 (ragged last batch, eager and HIP-graph replay paths
    mixed) and for a 2-shard split (what two ranks would score: shard_range blocks scored separately and concatenated -- the
    all-gather itself is covered by tests/test_distributed_cpu.py), so `update_X_L` selects identical images."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_POOL, SIZE = 200, 512
KW = dict(isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum', scaleUnc=False, showNMS=False, saveUnc=False,
          saveMaxConf=False, clsW=False)


class Loader:
    """what single_gpu_uncertainty reads from a DataLoader"""

    def __init__(self, ds, bs):
        self.dataset, self.batch_size, self.collate_fn = ds, bs, None


@pytest.fixture(scope='module')
def pool_model():
    import bench
    dev = torch.device('cuda', 0)
    model, _ = bench.build_model(dev, dict(bench.CONFIGS['voc512']))
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    cal = DevicePhiloxPool(4, (SIZE, SIZE), seed=21).device_batch([0, 1, 2, 3], dev)['img'][0].clone()
    bench.calibrate_head(model, cal)           # trained-like head: ~0.5 % of the anchors above the 0.3 foreground threshold
    return model.eval()


def test_pool_images_equal_the_numpy_philox_restatement():
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    from oracle.pool import philox_normal_image
    dev = torch.device('cuda', 0)
    ds = DevicePhiloxPool(1 << 40, (64, 96), seed=20)
    ids = [0, 7, 123456, (1 << 33) + 5]
    img = ds.device_batch(ids, dev)['img'][0].cpu().numpy()
    assert img.shape == (4, 3, 64, 96)
    for k, i in enumerate(ids):
        ref = philox_normal_image(20, i, 3 * 64 * 96).reshape(3, 64, 96)
        np.testing.assert_allclose(img[k], ref, rtol=2e-5, atol=2e-5)
    # a different seed / id is a different image; the same (seed, id) inside another batch is the same image, bit for bit
    again = ds.device_batch([5, 123456], dev)['img'][0].cpu().numpy()
    assert np.array_equal(again[1], img[2]) and not np.array_equal(again[0], img[0])
    other = DevicePhiloxPool(8, (64, 96), seed=21).device_batch([0], dev)['img'][0].cpu().numpy()
    assert not np.array_equal(other[0], img[0])
    full = np.concatenate([img[k].ravel() for k in range(4)])
    assert abs(full.mean()) < 0.02 and abs(full.std() - 1.0) < 0.02


@pytest.fixture(params=['bf16', 'bf16x3'])
def precision(request):
    """both arithmetic modes: in the reference-precision mode the batch-size independence also rests on the split-K rule (slices chosen from
    the per-image geometry, conv.hip choose_ksplit) and on the lattice / fused layer-1 launches"""
    from aod_meh_hua_amd import functional as AF
    AF.set_precision(request.param)
    yield request.param
    AF.set_precision(os.environ.get('AOD_CONV_PREC', 'bf16x3'))


def test_pool_scores_are_batching_and_sharding_invariant(pool_model, monkeypatch, precision):
    from aod_meh_hua_amd.apis import test as apis_test
    from aod_meh_hua_amd.apis.test import single_gpu_uncertainty
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    from aod_meh_hua_amd.parallel import shard_range
    from aod_meh_hua_amd.utils.active_datasets import update_X_L
    ds = DevicePhiloxPool(N_POOL, (SIZE, SIZE), seed=20)
    with torch.no_grad():
        u16 = single_gpu_uncertainty(pool_model, Loader(ds, 16), **KW).cpu().numpy()           # 12 full batches (graph replay) + one of 8
        u5 = single_gpu_uncertainty(pool_model, Loader(ds, 5), **KW).cpu().numpy()             # 40 batches of 5
        # what rank r of a 2-rank run scores: the loop's own shard_range block with GLOBAL image ids (the all-gather is replaced by
        # the identity here and covered by tests/test_distributed_cpu.py)
        parts = []
        monkeypatch.setattr(apis_test, 'gather_scores', lambda local, n_total: local)
        for r in range(2):
            monkeypatch.setattr(apis_test, 'get_dist_info', lambda r=r: (r, 2))
            parts.append(single_gpu_uncertainty(pool_model, Loader(ds, 16), **KW).cpu().numpy())
            lo, hi, _ = shard_range(N_POOL, r, 2)
            assert parts[-1].shape == (hi - lo,)
        # ... and of the INTERLEAVED partition (AOD_POOL_SHARD=interleaved: global batch k -> rank k mod 2)
        monkeypatch.setenv('AOD_POOL_SHARD', 'interleaved')
        monkeypatch.setattr(apis_test, 'gather_scores_indexed', lambda local, idx, n_total, per=None: (local, idx))
        inter = np.full(N_POOL, np.nan, np.float32)
        for r in range(2):
            monkeypatch.setattr(apis_test, 'get_dist_info', lambda r=r: (r, 2))
            v, idx = single_gpu_uncertainty(pool_model, Loader(ds, 16), **KW)
            assert idx[:16] == list(range(16 * r, 16 * r + 16))
            inter[np.array(idx)] = v.cpu().numpy()
        monkeypatch.undo()
    assert np.array_equal(inter, u16)
    assert u16.shape == (N_POOL,) and np.isfinite(u16).all()
    assert (u16 > 0).sum() >= N_POOL // 2, 'degenerate pool: the calibrated head should give most images a non-zero score'
    assert np.array_equal(u16, u5), np.abs(u16 - u5).max()
    assert np.array_equal(u16, np.concatenate(parts))
    sel = []
    for u in (u16, u5, np.concatenate(parts)):
        np.random.seed(20)
        sel.append(update_X_L(u.astype(np.float64), np.arange(N_POOL), np.arange(10), 20, zeroRate=0.15))
    for a, b in sel[1:]:
        assert np.array_equal(a, sel[0][0]) and np.array_equal(b, sel[0][1])
    # a deep copy scores identically (the captured scoring graph is per model, the Philox stream is keyed by the image id only)
    m2 = copy.deepcopy(pool_model)
    with torch.no_grad():
        u2 = single_gpu_uncertainty(m2, Loader(ds, 16), **KW).cpu().numpy()
    assert np.array_equal(u2, u16)


def test_deferred_scoring_survives_the_caller_recycling_its_inputs(pool_model, precision):
    """ADVICE r4 (graphs.py): with defer=True the selection half of batch k runs on a second stream a whole conv half later.  The caller's
    `image_ids` / `img` may be local tensors that are freed, re-allocated or overwritten in place right after the call -- the slot's copies
    must have been taken by then.  Every deferred score must equal the score of the same (image, id) from a plain, non-deferred call."""
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    from aod_meh_hua_amd.graphs import GraphedScore
    dev = torch.device('cuda', 0)
    ds = DevicePhiloxPool(64, (256, 256), seed=33)
    B, NB = 4, 6
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), ori_shape=(256, 256, 3), scale_factor=np.ones(4, np.float32), flip=False)
             for _ in range(B)]
    kw = dict(rescale=True, isEval=False, batchIdx=0, **KW)
    batches = [ds.device_batch(list(range(k * B, k * B + B)), dev)['img'][0].clone() for k in range(NB)]
    with torch.no_grad():
        gs = GraphedScore(pool_model, **kw)
        ref = []
        for k in range(NB):
            _, u = gs(batches[k], metas, torch.arange(k * B, k * B + B, device=dev))
            ref.append(u.clone())
        torch.cuda.synchronize()
        got = []
        buf = torch.empty_like(batches[0])
        for k in range(NB):
            ids = torch.arange(k * B, k * B + B, device=dev)
            buf.copy_(batches[k])
            _, u = gs(buf, metas, ids, defer=True)
            got.append(u)
            ids.fill_(10 ** 6 + k)                   # overwritten in place ...
            del ids
            junk = torch.full((B,), 7 + k, dtype=torch.int64, device=dev)      # ... and its block handed out again
            buf.normal_()                            # the image buffer refilled at once
            del junk
        gs.sync()
        torch.cuda.synchronize()
    for k in range(NB):
        assert torch.equal(got[k], ref[k]), (k, got[k], ref[k])
    assert float(torch.stack(ref).abs().sum()) > 0


def test_512_image_pool_in_both_partitions_matches_one_rank_and_the_oracle(pool_model, monkeypatch):
    """BASELINE configs[3] at a size the suite can afford (512 on-device images of 256 x 256 instead of 10 000 of 512 x 512; the 10 k run is
    `bench.py --mode pool`, profiles/r06_bench_pool10k.json): the score vector of the product's pool loop is the SAME BYTES when one rank scores
    the pool and when two ranks score it in the contiguous and in the interleaved partition (their shards concatenated / scattered like the
    all-gather does: tests/test_distributed_cpu.py covers the collective), and eight images spread over the pool agree with the CPU oracle run
    on the same weights, images and Philox stream (tolerances of tests/test_gpu_precision_end_to_end.py)."""
    import hashlib
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.apis import test as apis_test
    from aod_meh_hua_amd.apis.test import single_gpu_uncertainty
    from aod_meh_hua_amd.datasets import DevicePhiloxPool
    from aod_meh_hua_amd.parallel import shard_range
    from oracle import model as omodel
    AF.set_precision('bf16x3')
    N, S, bs = 512, 256, 16
    dev = torch.device('cuda', 0)
    ds = DevicePhiloxPool(N, (S, S), seed=20)
    digest = lambda u: hashlib.sha256(np.ascontiguousarray(u, np.float32).tobytes()).hexdigest()[:16]
    with torch.no_grad():
        one = single_gpu_uncertainty(pool_model, Loader(ds, bs), **KW).cpu().numpy()
        monkeypatch.setattr(apis_test, 'gather_scores', lambda local, n_total: local)
        parts = []
        for r in range(2):
            monkeypatch.setattr(apis_test, 'get_dist_info', lambda r=r: (r, 2))
            parts.append(single_gpu_uncertainty(pool_model, Loader(ds, bs), **KW).cpu().numpy())
            lo, hi, _ = shard_range(N, r, 2)
            assert parts[-1].shape == (hi - lo,)
        monkeypatch.setenv('AOD_POOL_SHARD', 'interleaved')
        monkeypatch.setattr(apis_test, 'gather_scores_indexed', lambda local, idx, n_total, per=None: (local, idx))
        inter = np.full(N, np.nan, np.float32)
        for r in range(2):
            monkeypatch.setattr(apis_test, 'get_dist_info', lambda r=r: (r, 2))
            v, idx = single_gpu_uncertainty(pool_model, Loader(ds, bs), **KW)
            inter[np.array(idx)] = v.cpu().numpy()
        monkeypatch.undo()
    assert one.shape == (N,) and np.isfinite(one).all() and (one > 0).sum() >= N // 2
    assert digest(one) == digest(np.concatenate(parts)) == digest(inter), (digest(one), digest(np.concatenate(parts)), digest(inter))
    # ---- eight images against the oracle
    ids = [0, 65, 130, 195, 260, 325, 390, 511]
    imgs = ds.device_batch(ids, dev)['img'][0].cpu()
    sd = {k: v.detach().float().cpu() for k, v in pool_model.state_dict().items()}
    metas = [dict(img_shape=(S, S, 3), scale_factor=np.ones(4, np.float32)) for _ in ids]
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    o = omodel.score_images(sd, imgs, [m['img_shape'] for m in metas], [m['scale_factor'] for m in metas], sampler='philox', seed=20,
                            image_ids=np.array(ids))
    ref, got = np.array(o['unc'], np.float64), one[ids].astype(np.float64)
    assert np.array_equal(ref == 0, got == 0), (ref, got)
    nz = ref > 0
    assert nz.sum() >= 4
    dev_ = np.abs(got[nz] - ref[nz]) / ref[nz]
    assert dev_.max() < 1e-2 and np.median(dev_) < 1e-3, dev_
