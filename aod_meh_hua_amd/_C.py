"""ctypes binding of libaodhip.so (the C ABI declared in include/aod_hip.h).

The product path has NO CPU fallback: importing this module without the built library, or
calling an op on a non-GPU tensor, raises.  Tensors are passed as raw device pointers; kernels
are enqueued on torch's current HIP stream."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AOD_HIP_LIB') or os.path.join(_HERE, 'lib', 'libaodhip.so')      # AOD_HIP_LIB: instrumented debug builds (tools/dbg)


class AodHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} is missing: run `python -m aod_meh_hua_amd.build` (hipcc, gfx950). '
                      'There is no CPU fallback for the MEH/HUA hot path.')
lib = C.CDLL(LIB_PATH)


class ConvSeg(C.Structure):
    _fields_ = [('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32),
                ('src_row0', C.c_int64), ('dst_row0', C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [('C', C.c_int32), ('N', C.c_int32), ('R', C.c_int32), ('S', C.c_int32), ('stride', C.c_int32),
                ('pad', C.c_int32), ('dil', C.c_int32), ('transposed', C.c_int32), ('relu', C.c_int32),
                ('out_f32', C.c_int32), ('nseg', C.c_int32), ('x3', C.c_int32), ('seg', ConvSeg * 8)]


P, I32, I64, F32, U64, SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t
_SIGS = {
    'aod_version': (C.c_int, []),
    'aod_conv_x3p_count': (C.c_int64, []),
    'aod_set_pointwise_mode': (C.c_int, [I32]),
    'aod_bottleneck64_fwd': (C.c_int, [P, I32, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck64x3_fwd': (C.c_int, [P, I32, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck64x3_ds_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck64_ds_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_halo_conv3x3_x3_applies': (C.c_int, [P]),
    'aod_halo_conv3x3_x3': (C.c_int, [P, P, P, P, P, P]),
    'aod_mfma_clock_probe': (C.c_int, [I32, I32, P, P, P]),
    'aod_bottleneck128x3_bwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck128x3_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck128_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_frag_pack': (C.c_int, [P, I32, I32, P]),
    'aod_frag_pack_item_bytes': (C.c_int, []),
    'aod_bottleneck256f_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck256f_bwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck_bwd': (C.c_int, [I32, P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_bottleneck256_fwd': (C.c_int, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_conv2d': (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, P, P, P]),
    'aod_conv2d_ws_bytes': (SZ, [C.POINTER(ConvDesc)]),
    'aod_conv2d_ws': (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    'aod_conv2d_grouped': (C.c_int, [C.POINTER(ConvDesc), I32, P, P, P, P, P, P, P]),
    'aod_halo_conv3x3_applies': (C.c_int, [C.POINTER(ConvDesc)]),
    'aod_halo_conv3x3': (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P, P, P]),
    'aod_conv2d_wgrad': (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P]),
    'aod_conv2d_wgrad_splits': (C.c_int, [C.POINTER(ConvDesc)]),
    'aod_conv2d_wgrad_slabs': (C.c_int, [C.POINTER(ConvDesc), P, P, P, I32, I64, P, P]),
    'aod_conv2d_wgrad_group_plan': (C.c_int, [P, I32, P]),
    'aod_conv2d_wgrad_grouped': (C.c_int, [P, I32, P, P, P, P, P, P, P]),
    'aod_unpack_wgrad_slabs_grouped': (C.c_int, [I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'aod_unpack_wgrad_slabs': (C.c_int, [P, I32, I64, P, I32, I32, I32, I32, I32, I32, P, P, P, P, P, P, P]),
    'aod_conv_row_table_bytes': (SZ, [C.POINTER(ConvDesc)]),
    'aod_conv_row_table': (C.c_int, [C.POINTER(ConvDesc), P, P]),
    'aod_pack_weight_fwd': (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
    'aod_param_prep': (C.c_int, [P, I32, I32, P]),
    'aod_param_prep_item_bytes': (C.c_int, []),
    'aod_pack_weight_dgrad': (C.c_int, [P, P, I32, I32, I32, I32, I32, P, P]),
    'aod_unpack_wgrad': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, I32, P, P, P, P, P, P, P]),
    'aod_nchw_f32_to_nhwc_bf16': (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
    'aod_nchw_f32_to_s2d_bf16': (C.c_int, [P, P, I32, I32, I32, I32, P]),
    'aod_stem_pool_fwd': (C.c_int, [P, P, P, P, P, I32, I32, I32, P]),
    'aod_set_deterministic': (C.c_int, [I32]),
    'aod_get_deterministic': (C.c_int, []),
    'aod_stem_pool_x3_fwd': (C.c_int, [P, P, P, P, P, I32, I32, I32, I32, P]),
    'aod_maxpool3x3s2': (C.c_int, [P, P, I32, I32, I32, I32, P]),
    'aod_upsample2x_add': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_upsample2x_add_bwd': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_upsample2x_add_to': (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_upsample2x_add_bwd_set': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_act_bwd': (C.c_int, [P, P, P, P, P, P, P, P, P, P, I64, I32, I32, I32, P]),
    'aod_add_relu': (C.c_int, [P, P, P, I64, P]),
    'aod_pad_cast_colsum': (C.c_int, [P, P, P, P, I64, I32, I32, I32, P]),
    'aod_assign_ws_bytes': (SZ, [I32, I32]),
    'aod_max_iou_assign': (C.c_int, [P, P, I64, I32, P, P, P, I32, F32, F32, F32, I32, I32, P, P, P, P, P, P, P, P, P, I32, P, P]),
    'aod_loss_partials_len': (SZ, [I64]),
    'aod_edl_focal_l1_fwd': (C.c_int, [P, P, P, P, P, P, I64, I32, F32, F32, P, P, P, P]),
    'aod_edl_focal_l1_bwd': (C.c_int, [P, P, P, P, P, P, I64, I32, F32, F32, P, P, P, F32, I32, P, P, I32, I32, I32, I32, P]),
    'aod_edl_focal_elem': (C.c_int, [P, P, I64, I32, F32, F32, P, P, P]),
    'aod_meh_loss_fwd': (C.c_int, [P, P, P, I64, P, P, P]),
    'aod_meh_loss_bwd': (C.c_int, [P, P, P, I64, P, P, I32, I32, I32, P]),
    'aod_loss_levels_partials_len': (SZ, [I32, P]),
    'aod_edl_focal_l1_levels_fwd': (C.c_int, [P, P, P, P, P, P, I32, P, I32, F32, F32, P, P, P, P, I32, P, P, P]),
    'aod_edl_focal_l1_levels_bwd': (C.c_int, [P, P, P, P, P, P, I32, P, I32, F32, F32, P, P, P, P, P, I32, I32, I32, I32, P]),
    'aod_meh_loss_levels_fwd': (C.c_int, [P, P, P, I32, P, P, P, P]),
    'aod_meh_loss_levels_bwd': (C.c_int, [P, P, P, I32, P, P, P, I32, I32, I32, P]),
    'aod_softmax_rowmax': (C.c_int, [P, I32, I64, I32, F32, P, P, I32, P]),
    'aod_topk_stable': (C.c_int, [P, I32, I64, I32, P, I64, P]),
    'aod_pre_nms_levels': (C.c_int, [I32, P, P, P, P, P, P, I32, I32, F32, I32, I32, P, P, P, P, F32, P, P, P, P, P, P, P, I64, P]),
    'aod_gather_decode': (C.c_int, [P, P, P, P, P, I32, I64, I32, I32, I64, P, P, P, P, F32, P, P, P, P, I64, I64, I64, I32, P]),
    'aod_nms_ws_bytes': (SZ, [I32, I32, I32]),
    'aod_multiclass_nms': (C.c_int, [P, P, I32, I32, I32, F32, F32, I32, P, P, P, P, P, P]),
    'aod_hua_ws_bytes': (SZ, [I32, I32]),
    'aod_hua_score': (C.c_int, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, F32, I32, U64, P, I32, I32, I32, P, P, I32, P, P, P]),
    'aod_maxpool_fwd': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    'aod_maxpool_bwd': (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    'aod_l2norm_fwd': (C.c_int, [P, P, P, I64, I32, F32, P]),
    'aod_l2norm_bwd': (C.c_int, [P, P, P, P, P, I64, I32, F32, P]),
    'aod_ssd_loss_fwd': (C.c_int, [P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, P, P, P, P]),
    'aod_ssd_loss_bwd': (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, P, P, P, P, P, P]),
    'aod_synth_normal_images': (C.c_int, [P, I32, I64, U64, P, P]),
    'aod_x3_split': (C.c_int, [P, P, I64, I32, P]),
    'aod_x3_merge': (C.c_int, [P, P, I64, I32, P]),
    'aod_x3_add': (C.c_int, [P, P, P, I64, P]),
    'aod_x3_nchw_f32_to_s2d': (C.c_int, [P, P, I32, I32, I32, I32, P]),
    'aod_x3_maxpool3x3s2': (C.c_int, [P, P, I32, I32, I32, I32, P]),
    'aod_x3_upsample2x_add_to': (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_x3_upsample2x_add_bwd_set': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, P]),
    'aod_x3_act_bwd': (C.c_int, [P, P, P, P, I64, I32, I32, P]),
    'aod_x3_pad_cast_colsum': (C.c_int, [P, P, P, P, I64, I32, P]),
    'aod_x3_nchw_f32_to_nhwc': (C.c_int, [P, P, I32, I32, I32, I32, P]),
    'aod_x3_maxpool_fwd': (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    'aod_x3_maxpool_bwd': (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    'aod_x3_l2norm_fwd': (C.c_int, [P, P, P, I64, I32, F32, P]),
    'aod_x3_l2norm_bwd': (C.c_int, [P, P, P, P, P, I64, I32, F32, P]),
    'aod_sgd_multi': (C.c_int, [P, P, P, P, I32, F32, P, F32, F32, I32, F32, P]),
}
for _n, (_r, _a) in _SIGS.items():
    if hasattr(lib, _n):
        getattr(lib, _n).restype = _r
        getattr(lib, _n).argtypes = _a
lib.aod_last_error.restype = C.c_char_p


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: no CPU path exists."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AodHipError('aod_meh_hua_amd ops need tensors on the MI355X (cuda:N); got a CPU tensor. There is no CPU fallback.')
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, name='aod'):
    if rc != 0:
        raise AodHipError(f'{name} failed ({rc}): {lib.aod_last_error().decode()}')


def call(name, *args):
    check(getattr(lib, name)(*args), name)
