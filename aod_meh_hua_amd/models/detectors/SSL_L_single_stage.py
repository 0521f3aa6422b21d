"""SSL_L_SingleStageDetector / SSL_L_RetinaNet / SSD_L_SingleStageDetector plugins
(mmdet/models/detectors/SSL_L_single_stage.py:10-98, SSL_L_retinanet.py, SSD_L_single_stage.py)."""
import warnings

from ...core.bbox import bbox2result
from ..builder import DETECTORS, build_backbone, build_head, build_neck
from .SSL_Lambda import SSLBase_L_Detector


@DETECTORS.register_module()
class SSL_L_SingleStageDetector(SSLBase_L_Detector):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        if pretrained:
            warnings.warn('DeprecationWarning: pretrained is deprecated, please use "init_cfg" instead')
            backbone.pretrained = pretrained
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def extract_feat(self, img):
        from ... import hipops as ho
        with ho.scope('backbone'):
            x = self.backbone(img)
        if self.with_neck:
            with ho.scope('neck'):
                x = self.neck(x)
        return x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, **kwargs):
        """SSL_L_single_stage.py:51-62 -> (losses, head_out, feat_out)."""
        super().forward_train(img, img_metas)
        x = self.extract_feat(img)
        losses, head_out = self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, **kwargs)
        feat_out = [i.detach() for i in x]
        return losses, head_out, feat_out

    def forward_train_L(self, loss, head_out, feat_out, **kwargs):
        return self.bbox_head.forward_train_L(loss, head_out, feat_out, **kwargs)

    def simple_test(self, img, img_metas, rescale=False, **kwargs):
        """SSL_L_single_stage.py:68-98."""
        feat = self.extract_feat(img)
        if kwargs['isEval']:
            _results_list = self.bbox_head.simple_test(feat, img_metas, rescale=rescale, **kwargs)
            results_list = _results_list[0] if kwargs.get('isUnc') else _results_list
            return [bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes) for det_bboxes, det_labels in results_list]
        results_list, *uncertainties = self.bbox_head.simple_test(feat, img_metas, rescale=rescale, **kwargs)
        if self.test_cfg.uncertainty_pool in ('Entropy_NoNMS', 'Entropy_ALL', 'Entropy_NMS'):
            return (results_list, *uncertainties)
        bbox_results = [bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes) for det_bboxes, det_labels in results_list]
        return bbox_results, uncertainties

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError('test-time augmentation is not on the MEH/HUA path')


@DETECTORS.register_module()
class SSL_L_RetinaNet(SSL_L_SingleStageDetector):
    """mmdet/models/detectors/SSL_L_retinanet.py:1-18."""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained, init_cfg)


@DETECTORS.register_module()
class SSD_L_SingleStageDetector(SSL_L_SingleStageDetector):
    """mmdet/models/detectors/SSD_L_single_stage.py:10-134 (same control flow as SSL_L_SingleStageDetector)."""
