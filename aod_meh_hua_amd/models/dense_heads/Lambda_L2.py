"""Lambda_L2Net: RetinaNet head + Model Evidence Head (MEH) + HUA scoring, with the reference's
interface (mmdet/models/dense_heads/Lambda_L2.py:19-749).

MI355X re-design:
  * the three towers (cls / reg / L: 4 x [3x3 conv 256->256 + ReLU], weights shared across the five
    pyramid levels) run LEVEL-BATCHED: one implicit-GEMM launch per conv covers all five levels
    (M = B*5456 rows at 512^2) instead of the reference's 5 x multi_apply launches per conv
    (forward_single, :85-103); bias + ReLU are fused in the epilogue;
  * prediction convs write fp32 NHWC, which IS the [B, H*W*A, C] layout the losses / scoring want,
    so `permute(0,2,3,1).reshape` (:114,120,266) costs nothing;
  * loss_single (:112-121) is one fused kernel per level (EDL softmax-focal + L1 + row sums);
  * scoring (:254-384, 489-619) = softmax/row-max, stable top-k, gather+decode, class-aware greedy NMS
    and the fused HUA kernel (Dirichlet sampling -> entropies -> (object, level, class) bins -> aggregate)."""
import numpy as np
import torch
import torch.nn as nn

from ... import functional as AF
from ...core.utils import multi_apply
from ...mmcv_lite import Conv2d, ConvModule, force_fp32
from ..builder import HEADS
from .L_anchor_head import L_AnchorHead


@HEADS.register_module()
class Lambda_L2Net(L_AnchorHead):
    def __init__(self, num_classes, in_channels, stacked_convs=4, conv_cfg=None, norm_cfg=None,
                 anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                                       strides=[8, 16, 32, 64, 128]),
                 init_cfg=dict(type='Normal', layer='Conv2d', std=0.01,
                               override=dict(type='Normal', name='retina_cls', std=0.01, bias_prob=0.01)), **kwargs):
        self.stacked_convs, self.conv_cfg, self.norm_cfg = stacked_convs, conv_cfg, norm_cfg
        self.isTrainD = False
        super().__init__(num_classes, in_channels, anchor_generator=anchor_generator, init_cfg=init_cfg, **kwargs)
        self.L_names = ['retina_L', 'L_convs']

    def _init_layers(self):
        """Lambda_L2.py:38-54 (same attribute names -> same state_dict keys)."""
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs, self.reg_convs, self.L_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            for tower in (self.cls_convs, self.reg_convs, self.L_convs):
                tower.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        self.retina_cls = Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 3, padding=1)
        self.retina_reg = Conv2d(self.feat_channels, self.num_anchors * 4, 3, padding=1)
        self.retina_L = Conv2d(self.feat_channels, self.num_anchors, 3, padding=1)

    # ------------------------------------------------------------------ forward
    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        outs = self.forward(x, **kwargs)
        loss_inputs = outs + (None, gt_bboxes, gt_labels, img_metas)
        return self.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore, **kwargs)

    def forward_train_L(self, prev_loss, head_out, x, **kwargs):
        L_scores = self.forward_L(x, head_out, **kwargs)
        return self.loss_L(L_scores, head_out, prev_loss, **kwargs)

    def forward(self, feats, **kwargs):
        """Lambda_L2.py:79-94, all levels per launch.  Returns (cls_scores[L], bbox_preds[L]) fp32 [B, A*C, h, w]."""
        feats = list(feats)
        cls_feat, reg_feat = (list(t) for t in zip(*[AF.fork(f, 2) for f in feats]))      # (every level feeds both towers)
        if len(self.cls_convs) == len(self.reg_convs) and all(m.with_activation for m in list(self.cls_convs) + list(self.reg_convs)):
            # the two towers advance together: one grouped launch per depth (functional.ConvPairFn)
            # training: the MEH tower's forward (forward_L below, on the same -- detached -- pyramid, run by train_step_L right after this pass)
            # rides in the same grouped launches; forward_L then only records its autograd nodes around the stored outputs
            import os
            ride = (torch.is_grad_enabled() and self.training and os.environ.get('AOD_MEH_RIDER', '1') != '0'
                    and len(self.L_convs) == len(self.cls_convs) and all(m.with_activation for m in self.L_convs)
                    and feats[0].dtype == torch.bfloat16)
            self._L_pre = None
            L_rows, L_outs = (AF.multi_rows([f.detach() for f in feats])[0] if ride else None), []
            for i, (cc, rc) in enumerate(zip(self.cls_convs, self.reg_convs)):          # every tower activation has exactly one consumer
                if L_rows is not None:
                    cls_feat, reg_feat, L_rows = AF.conv_pair_act(cls_feat, reg_feat, cc.conv, rc.conv, sole_consumer=i > 0,
                                                                  rider=(self.L_convs[i].conv, L_rows))
                    L_outs.append(L_rows)
                else:
                    cls_feat, reg_feat = AF.conv_pair_act(cls_feat, reg_feat, cc.conv, rc.conv, sole_consumer=i > 0)
            if ride and len(L_outs) == len(self.L_convs) and all(o is not None for o in L_outs):
                self._L_pre = (feats[0].data_ptr(), tuple(feats[0].shape), [m.conv.weight._version for m in self.L_convs], L_outs)
        else:
            for i, conv in enumerate(self.cls_convs):
                cls_feat = conv(cls_feat, sole_consumer=i > 0)
            for i, conv in enumerate(self.reg_convs):
                reg_feat = conv(reg_feat, sole_consumer=i > 0)
        return (self.retina_cls(cls_feat, out_f32=True, sole_consumer=len(self.cls_convs) > 0),
                self.retina_reg(reg_feat, out_f32=True, sole_consumer=len(self.reg_convs) > 0))

    def forward_L(self, feats, head_out=None, **kwargs):
        """Lambda_L2.py:82-83,96-103: MEH tower + retina_L + ReLU (fused)."""
        L_feat = list(feats)
        pre, self._L_pre = getattr(self, '_L_pre', None), None
        if pre is not None and not (torch.is_grad_enabled() and pre[0] == L_feat[0].data_ptr() and pre[1] == tuple(L_feat[0].shape)
                                    and pre[2] == [m.conv.weight._version for m in self.L_convs]):
            pre = None                  # another pyramid, or the tower's weights changed since forward(): compute it here
        for i, conv in enumerate(self.L_convs):
            L_feat = conv(L_feat, sole_consumer=i > 0, pre=pre[3][i] if pre is not None else None)
        return self.retina_L(L_feat, relu=True, out_f32=True, sole_consumer=len(self.L_convs) > 0)

    def forward_single(self, x):
        c, r = self.forward([x])
        return c[0], r[0]

    def forward_single_L(self, x):
        return self.forward_L([x])[0], 0

    # ------------------------------------------------------------------ losses
    _can_defer_avg = True

    @force_fp32(apply_to=('cls_score', 'bbox_pred'))
    def loss_single(self, cls_score, bbox_pred, anchors, labels, label_weights, bbox_targets, bbox_weights, sIdx, num_total_samples, **kwargs):
        """Live branch of Lambda_L2.py:112-121 (Labeled and not Pseudo).  The Pseudo branch (:122-232) is never
        entered by the AL driver (SURVEY 3.2 iv) and is not built."""
        if not (kwargs.get('Labeled', True) and not kwargs.get('Pseudo', False)):
            raise NotImplementedError('pseudo-label branch (Lambda_L2.py:122-232) is dead code in the reference driver')
        assert type(self.loss_bbox).__name__ == 'L1Loss' and type(self.loss_cls).__name__ == 'EDL_Softmax_FocalLoss'
        sum_cls, sum_box, loss_noR, sum_noR = AF.RetinaLossFn.apply(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
                                                                    float(self.loss_cls.gamma), float(self.loss_cls.alpha), self.cls_out_channels,
                                                                    kwargs.get('grad_arena'), int(sIdx))
        wc, wb = self.loss_cls.loss_weight, self.loss_bbox.loss_weight
        scaled = lambda w, t: t if w == 1.0 else w * t          # (1.0 * t == t exactly: no launch for the default weights)
        if kwargs.get('defer_avg'):                               # loss() divides all levels by their counts in one launch
            return scaled(wc, sum_cls), scaled(wb, sum_box), scaled(wc, loss_noR), scaled(wc, sum_noR)
        loss_cls = scaled(wc, sum_cls) / num_total_samples
        loss_bbox = scaled(wb, sum_box) / num_total_samples
        return loss_cls, loss_bbox, scaled(wc, loss_noR)

    def loss_all_levels(self, cls_scores, bbox_preds, labels_list, lw_list, bt_list, bw_list, num_pos=None, **kwargs):
        """loss_single (Lambda_L2.py:105-121) for every level in one launch per pass: (sums [3, L] = per-level (sum l*w, sum |d|*bw, sum l),
        [loss_noR rows per level]) or None when the levels are not adjacent row ranges / the loss weights are not 1 (then: per level).
        With num_pos (the assigner's per-image positive counts): (Q [3, L] = the per-level loss terms -- sums divided by num_total_samples /
        the level's row count --, rows, num_total_samples)."""
        if not (kwargs.get('Labeled', True) and not kwargs.get('Pseudo', False)):
            raise NotImplementedError('pseudo-label branch (Lambda_L2.py:122-232) is dead code in the reference driver')
        if self.loss_cls.loss_weight != 1.0 or self.loss_bbox.loss_weight != 1.0:
            return None
        assert type(self.loss_bbox).__name__ == 'L1Loss' and type(self.loss_cls).__name__ == 'EDL_Softmax_FocalLoss'
        C = self.cls_out_channels
        cls_d, box_d = AF.dense_levels(cls_scores), AF.dense_levels(bbox_preds)
        flat = [AF.dense_concat([t.reshape(-1, *t.shape[2:]) for t in ts]) for ts in (labels_list, lw_list, bt_list, bw_list)]
        if cls_d is None or box_d is None or any(f is None for f in flat) or cls_d.dtype != torch.float32 or box_d.dtype != torch.float32:
            return None
        A = cls_d.shape[1] // C
        level_rows = [c.shape[0] * c.shape[2] * c.shape[3] * A for c in cls_scores]
        out = AF.RetinaLossLevelsFn.apply(cls_d.view(-1, C), box_d.view(-1, 4), flat[0], flat[1], flat[2], flat[3],
                                          float(self.loss_cls.gamma), float(self.loss_cls.alpha), level_rows,
                                          (A, cls_d.shape[1], box_d.shape[1]), num_pos)
        return (out[0], list(out[1].split(level_rows))) + tuple(out[2:])

    def loss_all_levels_L(self, L_scores, losses, bw_list, **kwargs):
        """loss_single_L (Lambda_L2.py:235-241) for every level in one launch per pass: the per-level sums [L], or None."""
        lam_d = AF.dense_levels(L_scores)
        noR = AF.dense_concat([t.reshape(-1) for t in losses])
        bw = AF.dense_concat([t.reshape(-1, 4) for t in bw_list])
        if lam_d is None or noR is None or bw is None or lam_d.dtype != torch.float32 or noR.dtype != torch.float32 or noR.requires_grad:
            return None
        A = lam_d.shape[1]
        level_rows = [l.shape[0] * l.shape[2] * l.shape[3] * A for l in L_scores]
        if [int(t.numel()) for t in losses] != level_rows:
            return None
        return AF.MEHLossLevelsFn.apply(lam_d.view(-1), noR, bw, level_rows, A)

    @force_fp32(apply_to=('L_score'))
    def loss_single_L(self, L_score, loss, label_weights, bbox_weights, sIdx=0, **kwargs):
        """Lambda_L2.py:235-241: mean(((|lambda + 1e-9 - loss|) * bbox_weights[...,0])^2) * 5."""
        s = AF.MEHLossFn.apply(L_score, loss, bbox_weights, kwargs.get('grad_arena'), int(sIdx))
        if kwargs.get('defer_scale'):                             # loss_L() scales all levels in one launch
            return s, 5.0 / loss.numel()
        return s * (5.0 / loss.numel()), 0

    # ------------------------------------------------------------------ scoring
    def forward_all_towers(self, feats):
        """Scoring pass (no autograd): forward + forward_L (Lambda_L2.py:79-103) with the three towers advanced together -- the convs
        of one depth are ONE grouped launch (functional.conv_towers_nograd), whose 3 x 341 tiles fill whole rounds of the CUs."""
        feats = list(feats)
        c, r, l = feats, feats, feats
        for cc, rc, lc in zip(self.cls_convs, self.reg_convs, self.L_convs):
            c, r, l = AF.conv_towers_nograd([c, r, l], [cc.conv, rc.conv, lc.conv], relu=True)
        outs = (self.retina_cls(c, out_f32=True), self.retina_reg(r, out_f32=True))
        return outs, self.retina_L(l, relu=True, out_f32=True)

    def test_heads(self, feats):
        """the conv half of simple_test (Lambda_L2.py:398-403): ((cls_scores, bbox_preds), L_scores) of a scoring batch"""
        import os
        x3_ok = AF.get_precision() == 'bf16' or all(m.conv.weight.shape[0] % 256 == 0 and m.conv.weight.shape[1] % 32 == 0 for m in self.cls_convs)
        if (not torch.is_grad_enabled() and x3_ok and os.environ.get('AOD_GROUP_TOWERS', '1') != '0'
                and len(self.cls_convs) == len(self.reg_convs) == len(self.L_convs) > 0 and all(m.with_activation for m in self.cls_convs)):
            return self.forward_all_towers(feats)
        outs = self.forward(feats)
        return outs, self.forward_L(feats, head_out=None)

    def simple_test(self, feats, img_metas, rescale=False, _preds=None, **kwargs):
        """Lambda_L2.py:398-420.  `_preds` = the result of test_heads() computed earlier (graphs.GraphedScore runs the conv half and the
        selection / HUA half as two graphs on two streams: the half below keeps at most 16 workgroups busy)."""
        outs, L_scores = _preds if _preds is not None else self.test_heads(feats)
        if not kwargs['isEval'] and kwargs['uPool'] == 'Entropy_NoNMS':
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, with_nms=False, **kwargs)
        elif not kwargs['isEval'] and kwargs['uPool'] == 'Entropy_ALL':
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, with_nms=bool(kwargs.get('showNMS')), L_scores=L_scores, **kwargs)
        elif not kwargs['isEval'] and kwargs['uPool'] == 'Entropy_NMS':
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, with_nms=True, L_scores=L_scores, **kwargs)
        else:
            results_list = self.get_bboxes(*outs, img_metas, rescale=rescale, **kwargs)
        if not kwargs['isEval']:
            if kwargs.get('scaleUnc'):
                return results_list
            return (results_list[0], *results_list[1:])
        return results_list

    @force_fp32(apply_to=('mlvl_cls_scores', 'mlvl_bbox_preds', 'mlvl_anchors'))
    def _get_bboxes(self, mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, cfg, rescale=False, with_nms=True, **kwargs):
        """Lambda_L2.py:254-384 on the HIP scoring kernels (see ..scoring.ScoringPipeline)."""
        from ...scoring import score_batch
        cfg = self.test_cfg if cfg is None else cfg
        return score_batch(self, mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, cfg, rescale, with_nms, **kwargs)
