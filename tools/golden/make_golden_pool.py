"""Fixture for the selection-order parity test (reference path tools/train_RetinaNet.py:221-251 ->
mmdet/apis/test.py:90-135 -> mmdet/utils/active_datasets.py:102-135).

A pool of 64 planted-head images (tests/synth.planted_heads at 64 x 64, per-image seed; every 8th image has no planted object, so its score
is exactly 0 and the zeroRate branch of update_X_L is exercised) is scored by the ORACLE:
  * unc_philox  -- oracle/hua.py with the build's counter-based sampler (what the HIP kernels must reproduce to ~1e-6),
  * unc_torch   -- the same pipeline with torch.distributions.Dirichlet, i.e. what the reference itself computes (MC-500 noise), 3 reseeded runs,
and update_X_L (numpy seed 20, X_S_size 16, zeroRate 0.15) is applied to both.  The oracle takes ~4 s per image on a CPU core, which is
why its output is committed (tests/golden/pool_selection.npz) instead of being recomputed inside the GPU test; tests/test_oracle_golden.py
recomputes a few images to keep the fixture honest.

    python tools/golden/make_golden_pool.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as om          # noqa: E402
from oracle import selection as osel    # noqa: E402
from tests import synth                 # noqa: E402

N, H, W = 64, 64, 64


def pool_heads(i):
    return synth.planted_heads(1, H, W, seed=1000 + i, n_plant=0 if i % 8 == 0 else 3 + i % 4, plant_small=i % 8 != 0)


def score(i, sampler, seed=20):
    mt = synth.metas(1, H, W)
    if sampler == 'torch':
        torch.manual_seed(seed * 1000 + i)
    o = om.score_images(None, torch.zeros(1, 3, H, W), [m['img_shape'] for m in mt], [m['scale_factor'] for m in mt], sampler=sampler,
                        seed=20, heads=pool_heads(i), image_ids=[i])
    return float(o['unc'][0])


def select(unc):
    np.random.seed(20)
    return osel.update_X_L(np.asarray(unc, np.float64), np.arange(N), np.arange(8), 16, zeroRate=0.15)


if __name__ == '__main__':
    torch.set_num_threads(4)
    ph = np.array([score(i, 'philox') for i in range(N)])
    tr = np.stack([[score(i, 'torch', s) for i in range(N)] for s in (1, 2, 3)])
    XL, XU = select(ph)
    XLt = np.stack([select(t)[0] for t in tr])
    np.savez(os.path.join(ROOT, 'tests', 'golden', 'pool_selection.npz'), unc_philox=ph, unc_torch=tr, X_L_next=XL, X_U_next=XU, X_L_next_torch=XLt)
    print('zeros', int((ph == 0).sum()), 'X_L_next', XL, 'ref-sampler selections differ from philox by',
          [len(set(x) ^ set(XL)) for x in XLt])
