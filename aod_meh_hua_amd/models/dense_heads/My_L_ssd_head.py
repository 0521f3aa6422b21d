"""MyLSSDHead: SSD head + Model Evidence Head + HUA with the reference's interface
(mmdet/models/dense_heads/My_L_ssd_head.py:40-596 on top of My_anchor_head.py).  Per pyramid level one 3x3 conv for
classes (A*21, softmax with background), boxes (A*4) and lambda (A, ReLU).  Losses: softmax CE with 3:1 hard-negative mining +
SmoothL1(beta=1) per image in ONE fused kernel (`loss_single`, :182-215); MEH loss 2*mean((lambda+1e-9 - CE)^2) (:217-224,302-313).
Targets / scoring reuse the batch assign kernel and the scoring + HUA kernels (softmax-with-background mode, 21 Dirichlet columns)."""
import torch
import torch.nn as nn

from ... import functional as AF
from ...core.anchor import build_anchor_generator
from ...core.bbox import build_assigner, build_bbox_coder, build_sampler
from ...core.utils import multi_apply
from ...functional_ssd import SSDLossFn
from ...mmcv_lite import BaseModule, Conv2d, force_fp32
from ..builder import HEADS
from .L_anchor_head import L_AnchorHead


def _flat(x, c):
    """[B, A*c, h, w] channels_last -> [B, h*w*A, c] (view)."""
    B = x.shape[0]
    return x.permute(0, 2, 3, 1).reshape(B, -1, c)


@HEADS.register_module()
class MyLSSDHead(L_AnchorHead):
    def __init__(self, num_classes=80, in_channels=(512, 1024, 512, 256, 256, 256), stacked_convs=0, feat_channels=256, use_depthwise=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None,
                 anchor_generator=dict(type='SSDAnchorGenerator', scale_major=False, input_size=300, strides=[8, 16, 32, 64, 100, 300],
                                       ratios=([2], [2, 3], [2, 3], [2, 3], [2], [2]), basesize_ratio_range=(0.1, 0.9)),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                 reg_decoded_bbox=False, train_cfg=None, test_cfg=None,
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform', bias=0)):
        BaseModule.__init__(self, init_cfg)
        assert stacked_convs == 0 and not use_depthwise and not reg_decoded_bbox
        self.num_classes, self.in_channels, self.feat_channels = num_classes, in_channels, feat_channels
        self.last_activation = 'softmax'
        self.cls_out_channels = num_classes + 1
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors
        self._init_layers()
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.reg_decoded_bbox, self.sampling = False, False
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(dict(type='PseudoSampler'), context=self)
        self.fp16_enabled = False
        self.L_names = ['L_convs']

    def _init_layers(self):
        """My_L_ssd_head.py:102-137 (nn.Sequential wrappers keep the `cls_convs.{l}.0.weight` state_dict keys)."""
        self.cls_convs, self.reg_convs, self.L_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for channel, na in zip(self.in_channels, self.num_anchors):
            self.cls_convs.append(nn.Sequential(Conv2d(channel, na * self.cls_out_channels, kernel_size=3, padding=1)))
            self.reg_convs.append(nn.Sequential(Conv2d(channel, na * 4, kernel_size=3, padding=1)))
            self.L_convs.append(nn.Sequential(Conv2d(channel, na, kernel_size=3, padding=1)))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        outs = self.forward(x, **kwargs)
        return self.loss(*outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore, **kwargs)

    def forward_train_L(self, prev_loss, head_out, x, **kwargs):
        return self.loss_L(self.forward_L(x, head_out, **kwargs), head_out, prev_loss, **kwargs)

    def forward(self, feats, **kwargs):
        """:169-174."""
        from ... import functional as AF
        fa, fb = zip(*[AF.fork(f, 2) for f in feats])          # (every level feeds the cls and the reg conv)
        cls_scores = [c[0](f, out_f32=True) for f, c in zip(fa, self.cls_convs)]
        bbox_preds = [r[0](f, out_f32=True) for f, r in zip(fb, self.reg_convs)]
        return cls_scores, bbox_preds

    def forward_L(self, feats, head_out=None, **kwargs):
        """:176-180 (ReLU fused)."""
        return [l[0](f, relu=True, out_f32=True) for f, l in zip(feats, self.L_convs)]

    @force_fp32(apply_to=('cls_scores', 'bbox_preds'))
    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None, **kwargs):
        """:226-300.  Returns (dict(loss_cls [B x [1]], loss_bbox [B], loss_noR [B x [A]]), head_out)."""
        featmap_sizes = [tuple(f.shape[-2:]) for f in cls_scores]
        device = cls_scores[0].device
        labels_l, lw_l, bt_l, bw_l, num_total_pos, nla = self.get_targets_batch(featmap_sizes, img_metas, gt_bboxes, gt_labels, device)
        B = len(img_metas)
        C1 = self.cls_out_channels
        all_cls = torch.cat([_flat(s, C1) for s in cls_scores], 1)
        all_box = torch.cat([_flat(b, 4) for b in bbox_preds], 1)
        all_labels, all_lw = torch.cat(labels_l, 1), torch.cat(lw_l, 1)
        all_bt, all_bw = torch.cat(bt_l, 1), torch.cat(bw_l, 1)
        head_info = ['cls_scores', 'bbox_preds', 'all_anchor_list', 'labels_list', 'label_weights_list', 'bbox_targets_list',
                     'bbox_weights_list', 'num_total_samples']
        head_out = (head_info, cls_scores, bbox_preds, None, labels_l, lw_l, bt_l, bw_l, None)
        cls_sum, box_sum, ce = SSDLossFn.apply(all_cls, all_box, all_labels, all_lw, all_bt, all_bw, self.num_classes,
                                               int(self.train_cfg.neg_pos_ratio), float(self.train_cfg.smoothl1_beta))
        losses_cls = [(cls_sum[b] / num_total_pos)[None] for b in range(B)]
        losses_bbox = [box_sum[b] / num_total_pos for b in range(B)]
        losses_noR = [ce[b] for b in range(B)]
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox, loss_noR=losses_noR), head_out

    def loss_single_L(self, L_score, prev_loss, weight, **kwargs):
        """:217-224 (mineW is never passed by the driver)."""
        ones = torch.ones(L_score.numel(), 4, device=L_score.device)
        s = AF.MEHLossFn.apply(L_score.reshape(1, 1, -1, 1), prev_loss, ones)     # sum((l + 1e-9 - prev)^2)
        return s * (2.0 / L_score.numel()), 0

    def loss_L(self, L_scores, head_out, prev_loss, **kwargs):
        """:302-313 (the reference hard-codes batch 8 in a reshape; the batch size is taken from the tensors here)."""
        B = L_scores[0].shape[0]
        all_L = torch.cat([_flat(l, 1).reshape(B, -1) for l in L_scores], 1)
        loss_L = [self.loss_single_L(all_L[b], prev_loss[b], None)[0] for b in range(B)]
        return dict(loss_L=loss_L)

    # ------------------------------------------------------------------ scoring
    def simple_test(self, feats, img_metas, rescale=False, **kwargs):
        outs = self.forward(feats)
        L_scores = self.forward_L(feats, head_out=None)
        if not kwargs['isEval'] and kwargs['uPool'] in ('Entropy_ALL', 'Entropy_NMS'):
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, with_nms=kwargs['uPool'] == 'Entropy_NMS', L_scores=L_scores, **kwargs)
        else:
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, **kwargs)
        if not kwargs['isEval']:
            return (results_list[0], *results_list[1:])
        return results_list

    @force_fp32(apply_to=('mlvl_cls_scores', 'mlvl_bbox_preds', 'mlvl_anchors'))
    def _get_bboxes(self, mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, cfg, rescale=False, with_nms=True, **kwargs):
        """:315-433 on the HIP scoring kernels (softmax-with-background mode)."""
        from ...scoring import score_batch
        return score_batch(self, mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, self.test_cfg if cfg is None else cfg,
                           rescale, with_nms, **kwargs)
