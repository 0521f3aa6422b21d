from .backbones.resnet import ResNet
from .backbones.ssd_vgg import SSDVGG
from .builder import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone, build_detector, build_head, build_loss, build_neck)
from .dense_heads.L_anchor_head import L_AnchorHead
from .dense_heads.Lambda_L2 import Lambda_L2Net
from .dense_heads.My_L_ssd_head import MyLSSDHead
from .detectors.SSL_L_single_stage import SSD_L_SingleStageDetector, SSL_L_RetinaNet, SSL_L_SingleStageDetector
from .losses import EDL_Softmax_FocalLoss, L1Loss, SmoothL1Loss
from .necks.fpn import FPN
from .necks.ssd_neck import SSDNeck

__all__ = ['BACKBONES', 'NECKS', 'HEADS', 'LOSSES', 'DETECTORS', 'build_backbone', 'build_neck', 'build_head', 'build_loss',
           'build_detector', 'ResNet', 'FPN', 'L_AnchorHead', 'Lambda_L2Net', 'SSL_L_SingleStageDetector', 'SSL_L_RetinaNet',
           'EDL_Softmax_FocalLoss', 'L1Loss', 'SmoothL1Loss', 'SSDVGG', 'SSDNeck', 'MyLSSDHead', 'SSD_L_SingleStageDetector']
