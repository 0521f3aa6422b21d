# SSD512-VGG16 + MEH/HUA (reference: configs/ssd/ssd512_voc.py).  The reference's file is an override fragment (its `model`
# dict has neither type nor backbone): here it is written as what it overrides -- the SSD300 AL config -- plus exactly its overrides:
# input 512, a seventh pyramid level (neck out_channels / level_strides / level_paddings / last_kernel_size = 4), anchor generator with
# basesize_ratio_range (0.1, 0.9) and strides up to 512 (24 564 anchors per image).
_base_ = '../_base_/Config_SSD.py'
input_size = 512
model = dict(
    backbone=dict(input_size=input_size),
    neck=dict(out_channels=(512, 1024, 512, 256, 256, 256, 256), level_strides=(2, 2, 2, 2, 1), level_paddings=(1, 1, 1, 1, 1),
              last_kernel_size=4),
    bbox_head=dict(
        in_channels=(512, 1024, 512, 256, 256, 256, 256),
        anchor_generator=dict(type='SSDAnchorGenerator', scale_major=False, input_size=input_size, basesize_ratio_range=(0.1, 0.9),
                              strides=[8, 16, 32, 64, 128, 256, 512], ratios=[[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]])))
uncertainty_pool2 = 'objectSum_scaleAvg_classSum'
