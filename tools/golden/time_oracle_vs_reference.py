"""SURVEY 8(d) / BASELINE.md section 4 step 1: is the CPU oracle (oracle/model.py, what bench.py's `cpu_baseline` times on the GPU box)
a faithful proxy for "the reference's CPU path"?  Runs in the BUILD container only (the reference cannot travel): one training iteration
(train_step + backward, train_step_L + backward) of the REFERENCE -- imported from /root/reference under tools/golden/mmcv_shim.py -- and
of the oracle on identical seeded weights and inputs, B = 2 at 256 x 256, torch.set_num_threads(8).  Requires identical losses (bit for
bit in fp32) and reports the wall-time ratio (target: within ~5 %).

    python tools/golden/time_oracle_vs_reference.py            -> profiles/oracle_vs_reference_cpu.json
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')
import mmcv_shim  # noqa: E402

mmcv_shim.install()
from mmdet.models import build_detector  # noqa: E402

from oracle import model as om  # noqa: E402
from tests import synth  # noqa: E402

B, H, W, THREADS, REPS = 2, 256, 256, 8, 5


def main():
    torch.set_num_threads(THREADS)
    cfg, _ = mmcv_shim.load_reference_model_cfg('/root/reference/configs/_base_/Config_RetinaNet.py')
    model = build_detector(cfg)
    sd = om.seeded_state_dict(50, 20)
    model.load_state_dict(sd, strict=True)
    model.train()
    img = synth.images(B, H, W)
    gtb, gtl = synth.random_gts(B, H, W, seed=24, gmin=1, gmax=3)
    metas = synth.metas(B, H, W)

    def ref_iter():
        out, head_out, feat_out, prev = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), Labeled=True, Pseudo=False)
        model.zero_grad()
        out['loss'].backward()
        lossL = model.train_step_L(prev, head_out, feat_out)
        model.zero_grad()
        lossL['loss'].backward()
        return float(out['loss']), float(lossL['loss'])

    sdo = om.seeded_state_dict(50, 20)
    train_keys = [k for k, v in sdo.items() if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.'))]
    for k in train_keys:
        sdo[k].requires_grad_(True)

    def ora_iter():
        o = om.train_step(sdo, img, gtb, gtl)
        for k in train_keys:
            sdo[k].grad = None
        o['loss'].backward()
        oL = om.train_step_L(sdo, o['feats'], o['loss_noR'], o['targets'])
        for k in train_keys:
            sdo[k].grad = None
        oL['loss'].backward()
        return float(o['loss']), float(oL['loss'])

    def timed(fn):
        fn()                                    # warm-up (allocator, thread pool)
        ts, val = [], None
        for _ in range(REPS):
            t0 = time.perf_counter()
            val = fn()
            ts.append(time.perf_counter() - t0)
        return val, ts

    # interleave the two so that neither sees a systematically warmer machine
    (rl, rlL), tr1 = timed(ref_iter)
    (ol, olL), to1 = timed(ora_iter)
    _, tr2 = timed(ref_iter)
    _, to2 = timed(ora_iter)
    tr, to = sorted(tr1 + tr2), sorted(to1 + to2)
    med = lambda v: v[len(v) // 2]
    res = dict(what='one CPU training iteration (train_step + backward, train_step_L + backward), RetinaNet-R50-FPN + MEH, seeded weights',
               batch=B, size=[H, W], threads=THREADS, reps=2 * REPS,
               reference=dict(loss=rl, loss_L=rlL, median_s=round(med(tr), 4), min_s=round(tr[0], 4)),
               oracle=dict(loss=ol, loss_L=olL, median_s=round(med(to), 4), min_s=round(to[0], 4)),
               losses_bit_identical=bool(np.float32(rl) == np.float32(ol) and np.float32(rlL) == np.float32(olL)),
               wall_time_ratio_oracle_over_reference=round(med(to) / med(tr), 4))
    print(json.dumps(res, indent=1))
    with open(os.path.join(ROOT, 'profiles', 'oracle_vs_reference_cpu.json'), 'w') as f:
        json.dump(res, f, indent=1)
    assert res['losses_bit_identical'], 'the oracle no longer reproduces the reference train step'


if __name__ == '__main__':
    main()
