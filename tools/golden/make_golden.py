"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference under
tools/golden/mmcv_shim.py) on the seeded inputs of tests/synth.py.

Run in the build container only:   python tools/golden/make_golden.py
The fixtures are data (inputs are regenerated from seeds; outputs are stored); no reference
source or bytecode is copied.
"""
import hashlib
import os
import sys
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
from mmdet.core.anchor import AnchorGenerator  # noqa: E402
from mmdet.core.bbox.coder.delta_xywh_bbox_coder import bbox2delta, delta2bbox  # noqa: E402
from mmdet.core.bbox.assigners import MaxIoUAssigner  # noqa: E402
from mmdet.utils import active_datasets  # noqa: E402

from tests import synth  # noqa: E402
from oracle.model import seeded_state_dict  # noqa: E402  (weight RECIPE only; values go into the reference model)

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **kw):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **kw)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KB, {len(kw)} arrays')


torch.set_num_threads(8)
cfg, ns = mmcv_shim.load_reference_model_cfg('/root/reference/configs/_base_/Config_RetinaNet.py')
model = build_detector(cfg)
head = model.bbox_head

# ---------------------------------------------------------------- state_dict spec
sd_ref = model.state_dict()
save('state_dict_spec', keys=np.array(list(sd_ref.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd_ref.values()]),
     n_params=np.int64(sum(p.numel() for p in model.parameters())),
     n_trainable=np.int64(sum(p.numel() for p in model.parameters() if p.requires_grad)))

# ---------------------------------------------------------------- anchors / flags
ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
sizes_small = [(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)]
sizes_512 = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
g_small = ag.grid_anchors(sizes_small, device='cpu')
g_512 = ag.grid_anchors(sizes_512, device='cpu')
f_small = ag.valid_flags(sizes_small, (120, 96, 3), device='cpu')
f_512a = ag.valid_flags(sizes_512, (512, 512, 3), device='cpu')
f_512b = ag.valid_flags(sizes_512, (480, 500, 3), device='cpu')
save('anchors', base=np.stack([npy(b) for b in ag.base_anchors]),
     grid_small=np.concatenate([npy(a) for a in g_small]),
     grid512_sha=np.array([sha(a) for a in g_512]), grid512_first=np.stack([npy(a[:9]) for a in g_512]),
     grid512_last=np.stack([npy(a[-9:]) for a in g_512]),
     flags_small=np.concatenate([npy(f) for f in f_small]),
     flags512a_sum=np.array([int(f.sum()) for f in f_512a]), flags512b_sum=np.array([int(f.sum()) for f in f_512b]),
     flags512b_sha=np.array([sha(f) for f in f_512b]))

# ---------------------------------------------------------------- coder
g = synth.gen(23)
rois = torch.rand(64, 2, generator=g) * 100
rois = torch.cat([rois, rois + torch.rand(64, 2, generator=g) * 80 + 1], 1)
gts = torch.rand(64, 2, generator=g) * 100
gts = torch.cat([gts, gts + torch.rand(64, 2, generator=g) * 80 + 1], 1)
deltas = torch.randn(64, 4, generator=g) * torch.tensor([0.5, 0.5, 3.0, 3.0])
save('coder', rois=npy(rois), gts=npy(gts), deltas=npy(deltas), enc=npy(bbox2delta(rois, gts)),
     enc_ssd=npy(bbox2delta(rois, gts, stds=(.1, .1, .2, .2))),
     dec=npy(delta2bbox(rois, deltas, max_shape=(128, 160, 3))), dec_noclip=npy(delta2bbox(rois, deltas)),
     dec_batched=npy(delta2bbox(rois[None].expand(2, 64, 4), torch.stack([deltas, -deltas]),
                                max_shape=[(128, 160, 3), (90, 70, 3)])))

# ---------------------------------------------------------------- assignment / targets
H = W = 128
gtb, gtl = synth.assign_cases(H, W)
sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
anchor_list, valid_list = head.get_anchors(sizes, synth.metas(3, H, W), device='cpu')
tg = head.get_targets(anchor_list, valid_list, gtb, synth.metas(3, H, W), gt_bboxes_ignore_list=None,
                      gt_labels_list=gtl, label_channels=20)
labels_list, lw_list, bt_list, bw_list, num_pos, num_neg = tg
flat = torch.cat(anchor_list[0])
asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1)
gt_inds = [npy(asg.assign(flat, b, None, l).gt_inds) for b, l in zip(gtb, gtl)]
# SSD-style assigner config too (Config_SSD.py:56-62)
asg2 = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0., ignore_iof_thr=-1, gt_max_assign_all=False)
gt_inds2 = [npy(asg2.assign(flat, b, None, l).gt_inds) for b, l in zip(gtb, gtl)]
save('assign', gt_inds=np.stack(gt_inds), gt_inds_ssdcfg=np.stack(gt_inds2),
     labels=np.concatenate([npy(x) for x in labels_list], 1), label_weights=np.concatenate([npy(x) for x in lw_list], 1),
     bbox_targets=np.concatenate([npy(x) for x in bt_list], 1), bbox_weights=np.concatenate([npy(x) for x in bw_list], 1),
     num_total_pos=np.int64(num_pos), num_total_neg=np.int64(num_neg))
# docstring KAT (max_iou_assigner.py:86-91)
kat = MaxIoUAssigner(0.5, 0.5).assign(torch.Tensor([[0, 0, 10, 10], [10, 10, 20, 20]]), torch.Tensor([[0, 0, 10, 9]]))
assert kat.gt_inds.tolist() == [1, 0]

# ---------------------------------------------------------------- losses
li = synth.loss_inputs()
x = li['logits'].clone().requires_grad_(True)
bp = li['bbox_pred'].clone().requires_grad_(True)
n = li['num_total_samples']
loss_noR = head.loss_cls(x, li['labels'], reduction_override='none').sum(dim=-1)
loss_cls = head.loss_cls(x, li['labels'], li['label_weights'], avg_factor=n).sum()
loss_bbox = head.loss_bbox(bp, li['bbox_targets'], li['bbox_weights'], avg_factor=n)
total = loss_cls + loss_bbox + loss_noR.mean()
total.backward()
lam = li['lam'].clone().requires_grad_(True)
# loss_single_L wants L_score [B, A, h, w]; use B=1, A=1, h=N, w=1 so permute/reshape is the identity
loss_L, _ = head.loss_single_L(lam.view(1, 1, -1, 1), loss_noR.detach(), li['label_weights'].view(1, -1),
                               li['bbox_weights'].view(1, -1, 4))
loss_L.backward()
save('losses', loss_noR=npy(loss_noR), loss_cls=npy(loss_cls), loss_bbox=npy(loss_bbox), total=npy(total),
     grad_logits=npy(x.grad), grad_bbox=npy(bp.grad), loss_L=npy(loss_L), grad_lam=npy(lam.grad),
     in_sha=np.array([sha(li['logits']), sha(li['labels']), sha(li['bbox_pred'])]))

# full-model scoring at RANDOM init (before seeded weights are loaded): nothing exceeds 0.3 -> zeros
model.eval()
kw0 = dict(rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum',
           scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
with torch.no_grad():
    torch.manual_seed(0)
    img0 = synth.images(2, H, W)
    res = model(img=[img0], img_metas=[synth.metas(2, H, W)], return_loss=False, **kw0)
save('scoring_model', unc=np.array(res[1], dtype=np.float64), ndet=np.array([len(d[0]) for d in res[0]]))

# ---------------------------------------------------------------- tiny end-to-end train step
sd = seeded_state_dict(50, 20)
missing = model.load_state_dict(sd, strict=True)
model.train()
B = 2
img = synth.images(B, H, W)
gtb2, gtl2 = synth.random_gts(B, H, W, seed=24, gmin=1, gmax=3)
out, head_out, feat_out, prev = model.train_step(dict(img=img, img_metas=synth.metas(B, H, W), gt_bboxes=gtb2, gt_labels=gtl2),
                                                 Labeled=True, Pseudo=False)
model.zero_grad()
out['loss'].backward()
names = ['backbone.layer2.0.conv1.weight', 'backbone.layer2.0.bn1.weight', 'backbone.layer4.2.conv3.weight',
         'neck.lateral_convs.0.conv.weight', 'neck.fpn_convs.3.conv.weight', 'neck.fpn_convs.4.conv.bias',
         'bbox_head.cls_convs.0.conv.weight', 'bbox_head.reg_convs.3.conv.bias', 'bbox_head.retina_cls.weight',
         'bbox_head.retina_cls.bias', 'bbox_head.retina_reg.weight']
pd = dict(model.named_parameters())
gn_main = np.array([float(pd[k].grad.norm()) for k in names])
frozen_has_grad = any(pd[k].grad is not None for k in ('backbone.conv1.weight', 'backbone.layer1.0.conv1.weight'))
L_has_grad = pd['bbox_head.retina_L.weight'].grad is not None and float(pd['bbox_head.retina_L.weight'].grad.abs().sum()) > 0
lossL = model.train_step_L(prev, head_out, feat_out)
model.zero_grad()
lossL['loss'].backward()
namesL = ['bbox_head.L_convs.0.conv.weight', 'bbox_head.L_convs.3.conv.bias', 'bbox_head.retina_L.weight', 'bbox_head.retina_L.bias']
gn_L = np.array([float(pd[k].grad.norm()) for k in namesL])
with torch.no_grad():
    feats = model.extract_feat(img)
    cls_s, reg_s = head.forward(feats)
    L_s = head.forward_L(feats, None)
save('train_step', loss=npy(out['loss']), log_vars=np.array([out['log_vars'][k] for k in ('loss_cls', 'loss_bbox', 'loss_noR')]),
     loss_L=npy(lossL['loss']), grad_names=np.array(names), grad_norms=gn_main, grad_names_L=np.array(namesL), grad_norms_L=gn_L,
     frozen_has_grad=np.bool_(frozen_has_grad), L_has_grad_in_main=np.bool_(L_has_grad),
     num_total_samples=np.int64(head_out[8]),
     loss_noR_mean=np.array([float(p.mean()) for p in prev]), loss_noR_l4=npy(prev[4]), loss_noR_l2=npy(prev[2]),
     feat_absmean=np.array([float(f.abs().mean()) for f in feats]),
     feat_l4=npy(feats[4]), cls_l3=npy(cls_s[3]), reg_l4=npy(reg_s[4]), L_l3=npy(L_s[3]),
     cls_absmean=np.array([float(c.abs().mean()) for c in cls_s]))

# ---------------------------------------------------------------- scoring on planted logits
model.eval()
cls_p, reg_p, L_p = synth.planted_heads(B, H, W)
cap = {}
orig_cou = head.ComputeObjUnc


def spy(mlvl_cls_scores, pos_bboxes, mlvl_scores, mlvl_Ls, mlvl_idces):
    cap.update(pos=[p.clone() for p in pos_bboxes], scores=[s.clone() for s in mlvl_scores],
               Ls=[l.clone() for l in mlvl_Ls], idces=[i.clone() for i in mlvl_idces])
    o = orig_cou(mlvl_cls_scores, pos_bboxes, mlvl_scores, mlvl_Ls, mlvl_idces)
    cap['bins'] = o
    return o


head.ComputeObjUnc = spy
import mmdet.models.dense_heads.Lambda_L2 as L2mod  # noqa: E402
orig_nms = L2mod.multiclass_nms


def spy_nms(*a, **k):
    r = orig_nms(*a, **k)
    cap.setdefault('keep', []).append(r[2].clone())
    cap.setdefault('nms_in', []).append((a[0].clone(), a[1].clone()))
    return r


L2mod.multiclass_nms = spy_nms
kw = dict(rescale=True, with_nms=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS',
          uPool2='objectSum_scaleMax_classSum', scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False,
          clsW=False, batchIdx=0)
mt = synth.metas(B, H, W, scale=1.25)
uncs, bins_runs = [], []
with torch.no_grad():
    for seed in range(20):
        torch.manual_seed(seed)
        det_results, unc = head.get_bboxes(cls_p, reg_p, mt, L_scores=L_p, **kw)
        uncs.append(unc)
        flat_bins = {}
        for b, img_b in enumerate(cap['bins']):
            for o, obj in enumerate(img_b):
                for s, d in enumerate(obj):
                    for c, (ale, epi) in d.items():
                        flat_bins[(b, o, s, int(c))] = float(epi)
        bins_runs.append(flat_bins)
keys = sorted(bins_runs[0].keys())
assert all(sorted(r.keys()) == keys for r in bins_runs)
bin_vals = np.array([[r[k] for k in keys] for r in bins_runs])
dets = [npy(torch.cat([d, l[:, None].float()], 1)) for d, l in det_results]
save('scoring', unc_runs=np.array(uncs), bin_keys=np.array(keys), bin_vals=bin_vals,
     det0=dets[0], det1=dets[1], keep0=npy(cap['keep'][-2]), keep1=npy(cap['keep'][-1]),
     boxes_cat=npy(torch.stack([cap['nms_in'][-2][0], cap['nms_in'][-1][0]])), pos0=npy(cap['pos'][0]), pos1=npy(cap['pos'][1]),
     topk_idx=np.concatenate([npy(i) for i in cap['idces']], 1), lam=np.concatenate([npy(l) for l in cap['Ls']], 1),
     scores_sha=np.array([sha(s) for s in cap['scores']]), scores_l0_head=npy(cap['scores'][0][:, :8]),
     in_sha=np.array([sha(cls_p[0]), sha(reg_p[0]), sha(L_p[0])]))
head.ComputeObjUnc = orig_cou
L2mod.multiclass_nms = orig_nms

# ---------------------------------------------------------------- Entropy_ALL (ComputeScaleUnc / AggregateScaleUnc) on planted logits
cap2 = {}
orig_csu = head.ComputeScaleUnc


def spy2(mlvl_cls_scores, L_scores):
    o = orig_csu(mlvl_cls_scores, L_scores)
    cap2['bins'] = o
    return o


head.ComputeScaleUnc = spy2
kw_all = dict(kw, uPool='Entropy_ALL', uPool2='scaleAvg_classAvg', with_nms=False)
runs, binruns = [], []
with torch.no_grad():
    for seed in range(12):
        torch.manual_seed(seed)
        _, unc_all = head.get_bboxes(cls_p, reg_p, mt, L_scores=L_p, **kw_all)
        runs.append(unc_all)
        binruns.append({(b, s, int(c)): float(v[1]) for b, img_b in enumerate(cap2['bins']) for s, d in enumerate(img_b) for c, v in d.items()})
keys2 = sorted(binruns[0].keys())
assert all(sorted(r.keys()) == keys2 for r in binruns)
with torch.no_grad():
    torch.manual_seed(0)
    _, unc_ss = head.get_bboxes(cls_p, reg_p, mt, L_scores=L_p, **dict(kw_all, uPool2='scaleSum_classSum'))
save('scoring_all', unc_runs=np.array(runs, dtype=np.float64), bin_keys=np.array(keys2), bin_vals=np.array([[r[k] for k in keys2] for r in binruns]),
     unc_sumsum_seed0=np.array(unc_ss, dtype=np.float64))
head.ComputeScaleUnc = orig_csu

# ---------------------------------------------------------------- selection rule
rng_unc = np.random.RandomState(5)
unc = rng_unc.rand(400) * (rng_unc.rand(400) > 0.4)
X_all = np.arange(400)
X_L = np.sort(np.random.RandomState(6).choice(400, 40, replace=False))
np.random.seed(20)
XL1, XU1 = active_datasets.update_X_L(unc.copy(), X_all, X_L.copy(), X_S_size=20, zeroRate=0.15)
np.random.seed(20)
XL2, XU2 = active_datasets.update_X_L(unc.copy(), X_all, X_L.copy(), X_S_size=20)
save('selection', unc=unc, X_L=X_L, XL_zero=XL1, XU_zero=XU1, XL_plain=XL2, XU_plain=XU2)
print('done')
