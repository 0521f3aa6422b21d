from .backbones.resnet import ResNet
from .builder import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone, build_detector, build_head, build_loss, build_neck)
from .dense_heads.L_anchor_head import L_AnchorHead
from .dense_heads.Lambda_L2 import Lambda_L2Net
from .detectors.SSL_L_single_stage import SSL_L_RetinaNet, SSL_L_SingleStageDetector
from .losses import EDL_Softmax_FocalLoss, L1Loss, SmoothL1Loss
from .necks.fpn import FPN

__all__ = ['BACKBONES', 'NECKS', 'HEADS', 'LOSSES', 'DETECTORS', 'build_backbone', 'build_neck', 'build_head', 'build_loss',
           'build_detector', 'ResNet', 'FPN', 'L_AnchorHead', 'Lambda_L2Net', 'SSL_L_SingleStageDetector', 'SSL_L_RetinaNet',
           'EDL_Softmax_FocalLoss', 'L1Loss', 'SmoothL1Loss']
