from .edl_softmax_focal_loss import EDL_Softmax_FocalLoss
from .smooth_l1_loss import L1Loss, SmoothL1Loss

__all__ = ['EDL_Softmax_FocalLoss', 'L1Loss', 'SmoothL1Loss']
