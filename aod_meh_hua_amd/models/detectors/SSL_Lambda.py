"""SSLBase_L_Detector: detector base with the reference's interface (mmdet/models/detectors/SSL_Lambda.py:12-168):
forward dispatch, _parse_losses (sums EVERY key containing 'loss', incl. loss_noR -- SURVEY 9 item 1),
train_step -> (outputs, head_out, feat_out, loss_noR), train_step_L.

Difference by design: `log_vars` holds detached 0-d DEVICE tensors instead of `.item()` floats, so the
training loop never syncs the host with the MI355X; loggers call float() only when they print."""
from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import torch

from ...mmcv_lite import BaseModule, auto_fp16


class SSLBase_L_Detector(BaseModule, metaclass=ABCMeta):
    def __init__(self, init_cfg=None):
        super().__init__(init_cfg)
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_bbox(self):
        return hasattr(self, 'bbox_head') and self.bbox_head is not None

    @abstractmethod
    def extract_feat(self, imgs):
        pass

    def extract_feats(self, imgs):
        assert isinstance(imgs, list)
        return [self.extract_feat(img) for img in imgs]

    def forward_train(self, imgs, img_metas, **kwargs):
        batch_input_shape = tuple(imgs[0].size()[-2:])
        for img_meta in img_metas:
            img_meta['batch_input_shape'] = batch_input_shape

    @abstractmethod
    def simple_test(self, img, img_metas, **kwargs):
        pass

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        assert len(imgs) == 1, 'test-time augmentation is not on the MEH/HUA path'
        for img, img_meta in zip(imgs, img_metas):
            for img_id in range(len(img_meta)):
                img_meta[img_id]['batch_input_shape'] = tuple(img.size()[-2:])
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    @auto_fp16(apply_to=('img', ))
    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """SSL_Lambda.py:115-124."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs, _data=img[0], _meta=img_metas[0])

    def _parse_losses(self, losses, **kwargs):
        """SSL_Lambda.py:126-154."""
        log_vars = OrderedDict()
        # (.mean() of a 0-d tensor is the tensor itself, bit for bit; skipping it saves a reduce launch and its backward per level)
        mean = lambda t: t if t.dim() == 0 else t.mean()
        # functional.PackedLosses whose vectors are the rows of ONE matrix (the level-fused loss launch hands back loss_cls / loss_bbox / loss_noR
        # as a [3, L] tensor): one row-sum launch for all names, and -- when every loss term of this call is such a row -- the total as the sum of
        # the row sums: 2 launches instead of 5 here and 2 instead of 14 in backward (three select-backward fills and copies, adds, ...)
        groups = {}
        for loss_name, loss_value in losses.items():
            g = getattr(loss_value, 'group', None)
            if g is not None:
                groups.setdefault(id(g[0]), [g[0], {}])[1][loss_name] = g[1]
        row_sums, whole = {}, None
        for mat, rows in groups.values():
            if sorted(rows.values()) == list(range(mat.shape[0])):
                lv = mat.sum(1)
                for loss_name, r in rows.items():
                    row_sums[loss_name] = lv[r]
                if len(groups) == 1 and len(rows) == len(losses) and all('loss' in k for k in rows):
                    whole = lv.sum()
        for loss_name, loss_value in losses.items():
            if loss_name in row_sums:
                log_vars[loss_name] = row_sums[loss_name]
            elif isinstance(loss_value, torch.Tensor):
                log_vars[loss_name] = mean(loss_value)
            elif getattr(loss_value, 'packed', None) is not None:
                log_vars[loss_name] = loss_value.packed.sum()        # functional.PackedLosses: the per-level means in one vector
            elif isinstance(loss_value, list):
                loss_sum = None
                for _loss in loss_value:
                    if torch.is_tensor(_loss):
                        loss_sum = mean(_loss) if loss_sum is None else loss_sum + mean(_loss)
                log_vars[loss_name] = loss_sum if loss_sum is not None else torch.zeros((), device=kwargs.get('device'))
            else:
                raise TypeError(f'{loss_name} is not a tensor or list of tensors')
        terms = [_value for _key, _value in log_vars.items() if 'loss' in _key]
        if whole is not None:
            loss = whole
        else:
            loss = terms[0]
            for _value in terms[1:]:           # (sum() would start from the int 0: one more launch)
                loss = loss + _value
        for loss_name, loss_value in log_vars.items():
            log_vars[loss_name] = loss_value.detach()
        return loss, log_vars

    def train_step(self, data, **kwargs):
        """SSL_Lambda.py:156-162."""
        losses, head_out, feat_out = self(**data, **kwargs)
        loss_noR = [i.detach() for i in losses['loss_noR']]
        loss, log_vars = self._parse_losses(losses, device=data['img'].device)
        outputs = dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
        return outputs, head_out, feat_out, loss_noR

    def train_step_L(self, prev_loss, head_out, feat_out, **kwargs):
        """SSL_Lambda.py:164-168."""
        losses = self.forward_train_L(prev_loss, head_out, feat_out, **kwargs)
        loss, log_vars = self._parse_losses(losses, device=prev_loss[0].device)
        return dict(loss=loss, log_vars=log_vars, num_samples=2)

    def grad_segments(self, params):
        """`params` (one optimizer's parameter list) split into the groups whose gradients become final together in
        functional.backward_segments(): [everything behind no cut (neck, heads), deepest backbone stage, ..., shallowest].  A backbone without
        cut points (SSD's VGG) gives one group."""
        ids = {id(q) for q in params}
        groups = [[q for q in g if id(q) in ids] for g in (self.backbone.grad_segments() if hasattr(self.backbone, 'grad_segments') else [])]
        groups = [g for g in groups if g]
        behind = {id(q) for g in groups for q in g}
        return [[q for q in params if id(q) not in behind]] + groups

    def val_step(self, data, optimizer=None, **kwargs):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses[0] if isinstance(losses, tuple) else losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
