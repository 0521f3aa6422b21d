from .anchor import (ANCHOR_GENERATORS, PRIOR_GENERATORS, AnchorGenerator, SSDAnchorGenerator, anchor_inside_flags,
                     build_anchor_generator, build_prior_generator, images_to_levels)
from .bbox import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS, AssignResult, BboxOverlaps2D, DeltaXYWHBBoxCoder,
                   MaxIoUAssigner, PseudoSampler, bbox2delta, bbox2result, bbox_overlaps, build_assigner, build_bbox_coder,
                   build_iou_calculator, build_sampler, delta2bbox)
from .utils import multi_apply, reduce_mean, unmap
