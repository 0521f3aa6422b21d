"""Generate tests/golden/ssd_*.npz by running the REFERENCE SSD300-VGG16 + MEH/HUA model (imported from /root/reference under
tools/golden/mmcv_shim.py) on the seeded inputs of tests/synth.py.   Run in the build container only:
    python tools/golden/make_golden_ssd.py
Fixtures are data (inputs are regenerated from seeds; outputs are stored)."""
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

from tests import synth  # noqa: E402
from oracle.model_ssd import seeded_state_dict  # noqa: E402  (weight RECIPE only; values go into the reference model)

OUT = os.path.join(ROOT, 'tests', 'golden')


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **kw):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **kw)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KB, {len(kw)} arrays')


torch.set_num_threads(8)
cfg, ns = mmcv_shim.load_reference_model_cfg('/root/reference/configs/_base_/Config_SSD.py')
model = build_detector(cfg)
head = model.bbox_head
sd_ref = model.state_dict()
spec_keys = np.array(list(sd_ref.keys()))
spec_shapes = np.array([str(tuple(v.shape)) for v in sd_ref.values()])
anchors = head.anchor_generator.grid_anchors([(s, s) for s in synth.SSD_SIZES], device='cpu')
save('ssd_spec', keys=spec_keys, shapes=spec_shapes, n_params=np.int64(sum(p.numel() for p in model.parameters())),
     base_anchors=np.concatenate([npy(b) for b in head.anchor_generator.base_anchors]),
     anchors_l0_head=npy(anchors[0][:16]), anchors_l3=npy(anchors[3]), anchors_l5=npy(anchors[5]),
     num_base=np.array(head.anchor_generator.num_base_anchors))

# ---------------------------------------------------------------- train step (B = 8: loss_L hard-codes reshape(8, -1))
model.load_state_dict(seeded_state_dict(20), strict=True)
model.train()
B, H, W = 8, 300, 300
img = synth.images(B, H, W, seed=41)
gtb, gtl = synth.random_gts(B, H, W, seed=42, gmin=1, gmax=3)
out, head_out, feat_out, prev = model.train_step(dict(img=img, img_metas=synth.metas(B, H, W), gt_bboxes=gtb, gt_labels=gtl),
                                                 Labeled=True, Pseudo=False)
model.zero_grad()
out['loss'].backward()
names = ['backbone.features.0.weight', 'backbone.features.0.bias', 'backbone.features.10.weight', 'backbone.features.21.weight',
         'backbone.features.28.bias', 'backbone.features.31.weight', 'backbone.features.33.weight', 'neck.l2_norm.weight',
         'neck.extra_layers.0.0.conv.weight', 'neck.extra_layers.1.1.conv.weight', 'neck.extra_layers.3.1.conv.bias',
         'bbox_head.cls_convs.0.0.weight', 'bbox_head.cls_convs.1.0.bias', 'bbox_head.cls_convs.5.0.weight',
         'bbox_head.reg_convs.0.0.weight', 'bbox_head.reg_convs.3.0.bias']
pd = dict(model.named_parameters())
gn_main = np.array([float(pd[k].grad.norm()) for k in names])
L_has_grad = pd['bbox_head.L_convs.0.0.weight'].grad is not None and float(pd['bbox_head.L_convs.0.0.weight'].grad.abs().sum()) > 0
lossL = model.train_step_L(prev, head_out, feat_out)
model.zero_grad()
lossL['loss'].backward()
namesL = ['bbox_head.L_convs.0.0.weight', 'bbox_head.L_convs.1.0.bias', 'bbox_head.L_convs.3.0.weight', 'bbox_head.L_convs.5.0.bias']
gn_L = np.array([float(pd[k].grad.norm()) for k in namesL])
with torch.no_grad():
    feats = model.extract_feat(img)
    cls_s, reg_s = head.forward(feats)
    L_s = head.forward_L(feats, None)
labels_cat = torch.cat(head_out[4], 1)
save('ssd_train_step', loss=npy(out['loss']), log_vars=np.array([out['log_vars'][k] for k in ('loss_cls', 'loss_bbox', 'loss_noR')]),
     loss_L=npy(lossL['loss']), grad_names=np.array(names), grad_norms=gn_main, grad_names_L=np.array(namesL), grad_norms_L=gn_L,
     L_has_grad_in_main=np.bool_(L_has_grad), n_pos=np.array([int(((l >= 0) & (l < 20)).sum()) for l in labels_cat]),
     labels_sum=np.array([int(l.sum()) for l in labels_cat]), loss_noR_mean=np.array([float(p.mean()) for p in prev]),
     loss_noR_img0=npy(prev[0]), feat_absmean=np.array([float(f.abs().mean()) for f in feats]),
     feat_l0_sample=npy(feats[0][0, :8, :6, :6]), feat_l3=npy(feats[3][:2]), feat_l5=npy(feats[5]),
     cls_l2=npy(cls_s[2][:2]), reg_l4=npy(reg_s[4]), L_l3=npy(L_s[3]),
     cls_absmean=np.array([float(c.abs().mean()) for c in cls_s]))

# ---------------------------------------------------------------- scoring on planted logits
model.eval()
Bs = 2
cls_p, reg_p, L_p = synth.planted_heads_ssd(Bs)
cap = {}
orig_cou = head.ComputeObjUnc


def spy(mlvl_cls_scores, pos_bboxes, mlvl_scores, mlvl_Ls, mlvl_idces):
    cap.update(pos=[p.clone() for p in pos_bboxes], scores=[s.clone() for s in mlvl_scores], Ls=[l.clone() for l in mlvl_Ls])
    o = orig_cou(mlvl_cls_scores, pos_bboxes, mlvl_scores, mlvl_Ls, mlvl_idces)
    cap['bins'] = o
    return o


head.ComputeObjUnc = spy
import mmdet.models.dense_heads.My_L_ssd_head as SSDmod  # noqa: E402
orig_nms = SSDmod.multiclass_nms


def spy_nms(*a, **k):
    r = orig_nms(*a, **k)
    cap.setdefault('keep', []).append(r[2].clone())
    cap.setdefault('nms_in', []).append((a[0].clone(), a[1].clone()))
    return r


SSDmod.multiclass_nms = spy_nms
kw = dict(rescale=True, with_nms=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum',
          scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
mt = synth.metas(Bs, 300, 300, scale=1.25)
uncs, bins_runs = [], []
with torch.no_grad():
    for seed in range(16):
        torch.manual_seed(seed)
        det_results, unc = head.get_bboxes(cls_p, reg_p, mt, L_scores=L_p, **kw)
        uncs.append(unc)
        flat = {}
        for b, img_b in enumerate(cap['bins']):
            for o, obj in enumerate(img_b):
                for s, d in enumerate(obj):
                    for c, (ale, epi) in d.items():
                        flat[(b, o, s, int(c))] = float(epi)
        bins_runs.append(flat)
keys = sorted(bins_runs[0].keys())
assert all(sorted(r.keys()) == keys for r in bins_runs)
dets = [npy(torch.cat([d, l[:, None].float()], 1)) for d, l in det_results]
save('ssd_scoring', unc_runs=np.array(uncs), bin_keys=np.array(keys), bin_vals=np.array([[r[k] for k in keys] for r in bins_runs]),
     det0=dets[0], det1=dets[1], keep0=npy(cap['keep'][-2]), keep1=npy(cap['keep'][-1]),
     boxes_cat=npy(torch.stack([cap['nms_in'][-2][0], cap['nms_in'][-1][0]])), pos0=npy(cap['pos'][0]), pos1=npy(cap['pos'][1]),
     lam=np.concatenate([npy(l) for l in cap['Ls']], 1), scores_l0_head=npy(cap['scores'][0][:, :8]),
     scores_l2=npy(cap['scores'][2]))
print('done')
