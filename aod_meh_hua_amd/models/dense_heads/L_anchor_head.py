"""L_AnchorHead: anchor-based dense-head base with the reference's interface
(mmdet/models/dense_heads/L_anchor_head.py:36-356): builds coder / assigner / sampler / anchor generator /
losses from config, `get_anchors`, `get_targets`, `loss` (returns (dict, head_out)), `loss_L`, `get_bboxes`.

MI355X re-design of the target pipeline: anchors and valid flags are cached per shape; assignment +
pseudo-sampling + delta encoding + unmap + images_to_levels for ALL images of the batch is one call into
the HIP kernels (3 launches) that writes level-major outputs, instead of the reference's per-image Python
loop of small torch ops (`_get_targets_single`, :155-202)."""
import torch

from ...core.anchor import build_anchor_generator
from ...core.bbox import build_assigner, build_bbox_coder, build_sampler
from ...core.utils import multi_apply
from ...mmcv_lite import BaseModule, Conv2d, force_fp32
from ..builder import HEADS, build_loss

_NO_SAMPLING = ['EDL_Loss', 'EDL_Loss_2', 'EDL_Loss_3', 'EDL_Loss_BCE', 'FocalLoss', 'GHMC', 'QualityFocalLoss', 'EDL_FocalLoss',
                'EDL_BetaFocalLoss', 'EDL_FocalLoss_Dummy', 'EDL_Softmax_FocalLoss', 'EDL_Softmax_FocalLoss_Dummy',
                'EDL_Softmax_SL_FocalLoss', 'SSL_EDL_Softmax_FocalLoss']


class PackedGT(tuple):
    """(gts [B,Gmax,4] f32, counts [B] int32, labels [B,Gmax] int64) already on the device: what pack_gts produces.  Passed as
    `gt_bboxes` (with gt_labels=None) by callers that keep static input buffers (graphs.GraphedTrainStep)."""

    def __len__(self):          # len(gt_bboxes) == batch size in the callers
        return int(self[1].shape[0])


def pack_gts(gt_bboxes, gt_labels, device):
    """list of [G_i,4] / [G_i] -> ([B,Gmax,4] f32, [B] int32 counts, [B,Gmax] int64); sizes are host-known."""
    if isinstance(gt_bboxes, PackedGT):
        return tuple.__getitem__(gt_bboxes, 0), tuple.__getitem__(gt_bboxes, 1), tuple.__getitem__(gt_bboxes, 2)
    B = len(gt_bboxes)
    counts = [int(g.shape[0]) for g in gt_bboxes]
    gmax = max(max(counts), 1)
    gts = torch.zeros(B, gmax, 4, device=device)
    labs = torch.zeros(B, gmax, dtype=torch.long, device=device)
    if sum(counts) > 0:
        bi = torch.tensor([b for b, c in enumerate(counts) for _ in range(c)], dtype=torch.long)
        gi = torch.tensor([i for c in counts for i in range(c)], dtype=torch.long)
        bi, gi = bi.to(device, non_blocking=True), gi.to(device, non_blocking=True)
        # (lists that live on one device -- host tensors from a loader, as a rule -- are joined there and cross over in ONE copy each:
        # a `.to(device)` per image was 2 B copies of a few boxes, a queue entry each)
        def joined(ts, dt, shape):
            if len({t.device for t in ts}) == 1:
                return torch.cat([t.reshape(shape) for t in ts]).to(device=device, dtype=dt, non_blocking=True)
            return torch.cat([t.to(device).to(dt).reshape(shape) for t in ts])
        gts[bi, gi] = joined(gt_bboxes, torch.float32, (-1, 4))
        if gt_labels is not None:
            labs[bi, gi] = joined(gt_labels, torch.long, (-1,))
    return gts, torch.tensor(counts, dtype=torch.int32).to(device, non_blocking=True), labs


@HEADS.register_module()
class L_AnchorHead(BaseModule):
    def __init__(self, num_classes, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=(.0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False, loss_cls=dict(type='CrossEntropyLoss', last_activation='sigmoid', loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0), train_cfg=None, test_cfg=None,
                 init_cfg=dict(type='Normal', layer='Conv2d', std=0.01)):
        super().__init__(init_cfg)
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.last_activation = loss_cls.get('last_activation')
        self.sampling = loss_cls['type'] not in _NO_SAMPLING
        if self.last_activation in ('sigmoid', 'relu'):
            self.cls_out_channels = num_classes
        elif self.last_activation in ('softmax', 'EDL_BG'):
            self.cls_out_channels = num_classes + 1
        if self.cls_out_channels <= 0:
            raise ValueError(f'num_classes={num_classes} is too small')
        self.reg_decoded_bbox = reg_decoded_bbox
        assert not reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            sampler_cfg = self.train_cfg.sampler if (self.sampling and hasattr(self.train_cfg, 'sampler')) else dict(type='PseudoSampler')
            self.sampler = build_sampler(sampler_cfg, context=self)
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()

    def _init_layers(self):
        self.conv_cls = Conv2d(self.in_channels, self.num_anchors * self.cls_out_channels, 1)
        self.conv_reg = Conv2d(self.in_channels, self.num_anchors * 4, 1)

    def forward_single(self, x):
        return self.conv_cls(x, out_f32=True), self.conv_reg(x, out_f32=True)

    def forward(self, feats):
        return multi_apply(self.forward_single, feats)

    # ------------------------------------------------------------------ anchors / targets
    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """L_anchor_head.py:129-153 (cached; the per-image lists alias the same tensors)."""
        mlvl = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [mlvl for _ in range(len(img_metas))]
        valid_flag_list = [self.anchor_generator.valid_flags(featmap_sizes, m['pad_shape'], device) for m in img_metas]
        return anchor_list, valid_flag_list

    def get_targets_batch(self, featmap_sizes, img_metas, gt_bboxes, gt_labels, device, raw_num_pos=False):
        """get_targets + _get_targets_single (L_anchor_head.py:155-257) for the whole batch on the GPU.
        Returns per-level lists (labels [B,A_l], label_weights, bbox_targets [B,A_l,4], bbox_weights),
        num_total_pos (0-d device tensor = sum_b max(#pos_b, 1)), num_level_anchors."""
        assert self.train_cfg.allowed_border < 0, 'allowed_border >= 0 is not used by the AL configs'
        assert self.train_cfg.get('pos_weight', -1) <= 0
        ag = self.anchor_generator
        flat = ag.flat_grid_anchors(featmap_sizes, device)
        nla = [a.shape[0] for a in ag.grid_anchors(featmap_sizes, device)]
        B = len(img_metas)
        pad_shapes = [tuple(int(v) for v in m['pad_shape'][:2]) for m in img_metas]
        flags = [ag.flat_valid_flags(featmap_sizes, ps, device) for ps in pad_shapes]
        key = ('allv', tuple(pad_shapes), tuple(map(tuple, featmap_sizes)))
        cache = ag._cache
        if key not in cache:
            cache[key] = all(bool(f.all()) for f in flags)          # one-time host check per shape
        valid = None if cache[key] else torch.stack(flags).to(torch.uint8)
        static = getattr(gt_bboxes, 'static', None)
        if static is not None:
            # HIP-graph replay (graphs.GraphedTrainStep): the flags live in a STATIC device buffer that is refreshed from each batch's
            # pad shapes before a replay, so one captured graph serves every batch of this tensor shape (keep-ratio VOC batches differ
            # in their per-image pad shapes, not in the padded batch shape)
            if static.get('valid') is None:
                static['valid'] = torch.stack(flags).to(torch.uint8).contiguous()
                stacked = {}

                def valid_fn(metas):
                    """(flags [B, anchors] uint8 on the device, key): one stacked tensor per combination of per-image pad shapes, built once --
                    a batch whose key equals the previous one's needs no refresh at all (graphs._load)"""
                    key = tuple(tuple(int(v) for v in m['pad_shape'][:2]) for m in metas)
                    t = stacked.get(key)
                    if t is None:
                        if len(stacked) >= 64:
                            stacked.clear()
                        t = stacked[key] = torch.stack([ag.flat_valid_flags(featmap_sizes, ps, device) for ps in key]).to(torch.uint8)
                    return t, key
                static['valid_fn'] = valid_fn
            valid = static['valid']
        gts, counts, labs = pack_gts(gt_bboxes, gt_labels, device)
        starts = [0]
        for n in nla:
            starts.append(starts[-1] + n)
        assigned, labels, lw, bt, bw, num_pos = self._assign(flat, valid, gts, counts, labs, starts)
        labels_l, lw_l, bt_l, bw_l = [], [], [], []
        lab_f, lw_f, bt_f, bw_f = labels.view(-1), lw.view(-1), bt.view(-1, 4), bw.view(-1, 4)
        for l, n in enumerate(nla):
            s, e = starts[l] * B, starts[l + 1] * B
            labels_l.append(lab_f[s:e].view(B, n)), lw_l.append(lw_f[s:e].view(B, n))
            bt_l.append(bt_f[s:e].view(B, n, 4)), bw_l.append(bw_f[s:e].view(B, n, 4))
        if raw_num_pos:         # (the level-fused loss launch forms num_total_samples itself)
            return labels_l, lw_l, bt_l, bw_l, num_pos, nla
        num_total_pos = num_pos.clamp(min=1).sum().float()
        return labels_l, lw_l, bt_l, bw_l, num_total_pos, nla

    def _assign(self, flat, valid, gts, counts, labs, starts):
        from ... import hipops as ho
        a = self.assigner
        return ho.max_iou_assign(flat, valid, gts, counts, labs, float(a.pos_iou_thr), float(a.neg_iou_thr), float(a.min_pos_iou),
                                 bool(a.gt_max_assign_all), self.num_classes, tuple(self.bbox_coder.means), tuple(self.bbox_coder.stds),
                                 level_start=starts)

    # ------------------------------------------------------------------ losses
    @force_fp32(apply_to=('cls_scores', 'bbox_preds', 'd_scores'))
    def loss(self, cls_scores, bbox_preds, D_scores, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None, **kwargs):
        """L_anchor_head.py:290-320: returns (dict(loss_cls, loss_bbox, loss_noR), head_out)."""
        featmap_sizes = [tuple(f.shape[-2:]) for f in cls_scores]
        assert len(featmap_sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        want_fused = False
        if self._can_defer_avg:
            from ... import functional as AF
            want_fused = AF.LOSS_LEVELS
        labels_list, lw_list, bt_list, bw_list, num_pos, nla = self.get_targets_batch(featmap_sizes, img_metas, gt_bboxes, gt_labels, device,
                                                                                      raw_num_pos=True)
        assert not self.sampling
        fused = None
        if want_fused:                      # (heads on the HIP loss kernels)
            # every level in ONE launch per pass when the levels are adjacent row ranges of their buffers (they are: level-batched prediction
            # convs, level-major targets); the per-level loss terms [3, L] -- already divided by num_total_samples / the level's row count --
            # and the per-level loss_noR rows come back bit-identical to the per-level launches followed by the tensor divisions
            fused = self.loss_all_levels(cls_scores, bbox_preds, labels_list, lw_list, bt_list, bw_list, num_pos=num_pos, **kwargs)
        num_total_samples = fused[2] if fused is not None else num_pos.clamp(min=1).sum().float()
        B = cls_scores[0].shape[0]
        mlvl = self.anchor_generator.grid_anchors(featmap_sizes, device)
        all_anchor_list = [a[None].expand(B, a.shape[0], 4) for a in mlvl]
        head_info = ['cls_scores', 'bbox_preds', 'all_anchor_list', 'labels_list', 'label_weights_list', 'bbox_targets_list',
                     'bbox_weights_list', 'num_total_samples']
        head_out = (head_info, cls_scores, bbox_preds, all_anchor_list, labels_list, lw_list, bt_list, bw_list, num_total_samples)
        if self._can_defer_avg and fused is None:               # one gradient buffer per prediction conv, a row range per level
            kwargs = dict(kwargs, grad_arena=AF.GradArena([c.shape[0] * c.shape[2] * c.shape[3] for c in cls_scores]))
        if fused is None:
            outs = multi_apply(self.loss_single, cls_scores, bbox_preds, all_anchor_list, labels_list, lw_list, bt_list, bw_list,
                               list(range(len(cls_scores))), num_total_samples=num_total_samples, featmap_sizes=featmap_sizes,
                               defer_avg=self._can_defer_avg, **kwargs)
            losses_cls, losses_bbox, losses_noR = outs[0], outs[1], outs[2]
        if self._can_defer_avg:
            # the per-level "sum / num_total_samples" of loss_single (L_anchor_head.py:266-288) and the "mean(loss_noR)" of
            # _parse_losses (SSL_Lambda.py:136-141) for all levels at once: the same quotients from three launches (stack, divisor, divide)
            # instead of ten divisions and five row reductions -- and as many fewer in backward
            from ... import functional as AF
            if fused is not None:
                Q, losses_noR = fused[0], fused[1]
                return dict(loss_cls=AF.PackedLosses(Q[0].unbind(0), Q[0], group=(Q, 0)), loss_bbox=AF.PackedLosses(Q[1].unbind(0), Q[1], group=(Q, 1)),
                            loss_noR=AF.PackedLosses(losses_noR, Q[2], group=(Q, 2))), head_out
            L = len(losses_cls)
            S = torch.stack(list(losses_cls) + list(losses_bbox) + list(outs[3])).view(3, L)
            counts = self._level_counts([int(t.numel()) for t in losses_noR], S.device)
            D = torch.cat([num_total_samples.reshape(1).expand(2 * L), counts]).view(3, L)
            Q = S / D
            losses_cls = AF.PackedLosses(Q[0].unbind(0), Q[0])
            losses_bbox = AF.PackedLosses(Q[1].unbind(0), Q[1])
            losses_noR = AF.PackedLosses(losses_noR, Q[2])           # the rows stay what train_step hands to the MEH step
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox, loss_noR=losses_noR), head_out

    def _level_counts(self, counts, device):
        key = (tuple(counts), str(device))
        cache = self.__dict__.setdefault('_count_cache', {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = torch.tensor(counts, dtype=torch.float32, device=device)
        return cache[key]

    _can_defer_avg = False      # set by heads whose loss_single understands defer_avg

    def loss_all_levels(self, *args, **kwargs):     # heads with a level-fused loss launch override these; None = the per-level path
        return None

    def loss_all_levels_L(self, *args, **kwargs):
        return None

    @force_fp32(apply_to=('L_scores'))
    def loss_L(self, L_scores, head_out, losses, **kwargs):
        """L_anchor_head.py:322-327."""
        if self._can_defer_avg:
            from ... import functional as AF
            fused = self.loss_all_levels_L(L_scores, losses, head_out[7], **kwargs) if AF.LOSS_LEVELS else None
            if fused is not None:
                Q = fused * self._level_counts([5.0 / int(t.numel()) for t in losses], fused.device)     # 5 * mean(.) per level
                return dict(loss_L=AF.PackedLosses(Q.unbind(0), Q))
            arena = AF.GradArena([l.shape[0] * l.shape[2] * l.shape[3] for l in L_scores])
            sums, scales = multi_apply(self.loss_single_L, L_scores, losses, head_out[5], head_out[7], list(range(len(L_scores))),
                                       defer_scale=True, grad_arena=arena, **kwargs)
            Q = torch.stack(list(sums)) * self._level_counts([float(v) for v in scales], sums[0].device)     # 5 * mean(.) per level
            return dict(loss_L=AF.PackedLosses(Q.unbind(0), Q))
        losses_L, _ = multi_apply(self.loss_single_L, L_scores, losses, head_out[5], head_out[7], **kwargs)
        return dict(loss_L=losses_L)

    # ------------------------------------------------------------------ scoring
    @force_fp32(apply_to=('cls_scores', 'bbox_preds'))
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True, **kwargs):
        """L_anchor_head.py:329-356."""
        assert len(cls_scores) == len(bbox_preds)
        device = cls_scores[0].device
        featmap_sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        mlvl_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        mlvl_cls_scores = [c.detach() for c in cls_scores]
        mlvl_bbox_preds = [b.detach() for b in bbox_preds]
        B = cls_scores[0].shape[0]
        img_shapes = [img_metas[i]['img_shape'] for i in range(B)]
        scale_factors = [img_metas[i]['scale_factor'] for i in range(B)]
        return self._get_bboxes(mlvl_cls_scores, mlvl_bbox_preds, mlvl_anchors, img_shapes, scale_factors, cfg, rescale, with_nms, **kwargs)
