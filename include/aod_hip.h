/* libaodhip.so -- C ABI of the MI355X (gfx950) kernels behind the MEH/HUA hot path.
 *
 * The reference (MoonLab-YH/AOD_MEH_HUA) is pure Python; its native work is dispatched to
 * third-party libraries (cuDNN via torch, mmcv-full 1.3.8 CUDA ops, torch.distributions).
 * Each entry point below names the reference call site (file:line under /root/reference)
 * whose native dispatch it replaces.  INTEGRATION.md shows the Python (ctypes) binding a
 * maintainer of the reference would add at each site.
 *
 * Conventions (every entry point):
 *   - returns 0 on success, -1 bad argument/shape, -2 workspace too small, -3 HIP error
 *     (message via aod_last_error());
 *   - caller owns all memory (device pointers, pre-allocated outputs); no internal
 *     allocation, no host sync; kernels are enqueued on `stream` (a hipStream_t);
 *   - activations are NHWC ("channels-last") bf16 unless stated; weights are handed over in
 *     the reference's OIHW fp32 master layout and re-packed by aod_pack_weight_*;
 *   - int64 label / index tensors keep the reference's dtypes.
 */
#ifndef AOD_HIP_H
#define AOD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* aod_stream_t; /* hipStream_t */

const char* aod_last_error(void);
int aod_version(void);

/* ------------------------------------------------------------------ convolution (K1-K3, K5)
 * One pyramid segment = one [B, H, W, C] NHWC block inside a flat [rows, C] buffer.  A launch may
 * cover up to 8 segments that share weights (the five FPN levels of the head towers,
 * mmdet/models/dense_heads/Lambda_L2.py:79-103 multi_apply over levels). */
typedef struct {
  int32_t B, H, W;        /* source block geometry (input of fwd; dZ of dgrad)                */
  int32_t OH, OW;         /* destination block geometry (output of fwd; dX of dgrad)          */
  int64_t src_row0;       /* first row of this block in the source buffer                     */
  int64_t dst_row0;       /* first row of this block in the destination buffer                */
} aod_conv_seg_t;

typedef struct {
  int32_t C, N;           /* GEMM K-side channels (source channels), output channels          */
  int32_t R, S;           /* filter taps                                                      */
  int32_t stride, pad, dil;
  int32_t transposed;     /* 0: y = conv(x, w).  1: dgrad (dX = conv_transpose(dZ, w)); `w`
                             must then be the [C_in][R][S][C_out] packing (aod_pack_weight_dgrad) */
  int32_t relu;           /* apply max(v, 0) last                                             */
  int32_t out_f32;        /* destination dtype: 0 bf16, 1 fp32                                */
  int32_t nseg;
  int32_t x3;             /* 1: reference-precision mode.  bf16 operands are in the X-LAYOUT: a tensor of c logical channels has
                             2*ceil32(c) bf16 columns [h(0..31) | l(0..31) | h(32..63) | ...], value = head + tail (h = bf16(v),
                             l = bf16(v - h)); three MFMAs per 32 channels (xh*wh + xl*wh + xh*wl) give fp32-grade products.  C is then
                             the PHYSICAL source width (multiple of 64), N the LOGICAL output channel count; a bf16 destination,
                             res and mask are X-layout rows of 2*ceil32(N) columns, an fp32 destination has N columns.  (Fills the
                             padding the struct had in front of seg[]: the layout of the other fields is unchanged.  aod_version() >= 2.
                             Descriptors MUST be zero-initialised (memset / = {0}); any value other than 0 or 1 is rejected.) */
  aod_conv_seg_t seg[8];
} aod_conv_desc_t;

/* replaces: F.conv2d dispatched by mmcv ConvModule / nn.Conv2d in
 *   mmdet/models/backbones/resnet.py:165-205,262-301,598-610 (+ frozen-stat BN :647-656 folded
 *   into pre_scale/pre_shift), mmdet/models/necks/fpn.py:156-202,
 *   mmdet/models/dense_heads/Lambda_L2.py:85-103.
 * v = acc * pre_scale[n] + pre_shift[n] + res[m][n]; if (mask) v = mask[m][n] > 0 ? v : 0;
 * v *= post_scale[n]; if (relu) v = max(v, 0); dst[m][n] = v; optional zraw[m][n] = acc (bf16);
 * optional colsum[n] += sum_m v (fp32 atomics).  In a dgrad launch res / mask / colsum fuse the ACTIVATION backward
 * of the layer below into the epilogue: dst = (dX + res) * [mask > 0] is the masked gradient w.r.t. that layer's
 * pre-activation output and colsum its bias / BN-shift gradient (the autograd chain of Lambda_L2.py:85-94 and
 * resnet.py:262-301 without a separate elementwise pass).
 * Any of pre_scale/pre_shift/res/mask/post_scale/zraw/colsum may be NULL. */
int aod_conv2d(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst,
               const float* pre_scale, const float* pre_shift, const void* res, const void* mask,
               const float* post_scale, void* zraw, float* colsum, aod_stream_t stream);

/* Same operation with a caller-provided scratch buffer, which lets small-output / deep-reduction convolutions (at most 16 x 16
 * outputs per image and K >= 4096: the stride-2 3x3 on C5 that makes pyramid level P6, fpn.py:156-202 -- 1 024 output pixels,
 * K = 18 432 -- and the 3x3 convs of the last backbone stage, resnet.py:262-301) run split-K: the K-steps of a tile are divided over
 * several workgroups, each stores its fp32 partial tile into its own slab of `workspace` ([slices][M][N] fp32, need not be
 * initialised) and a second small kernel adds the slabs IN ORDER and applies the epilogue above -- deterministic, no atomics.
 * Whether and how K is sliced depends on K and on the per-image output size only, not on the batch: the summation order of an output
 * element never changes with the batching of the pool.
 * aod_conv2d_ws_bytes() is the size the launch heuristic wants for this descriptor (0: the direct kernel is used and workspace may
 * be NULL).  aod_conv2d() == aod_conv2d_ws() with workspace NULL. */
size_t aod_conv2d_ws_bytes(const aod_conv_desc_t* desc);
int aod_conv2d_ws(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst,
                  const float* pre_scale, const float* pre_shift, const void* res, const void* mask,
                  const float* post_scale, void* zraw, float* colsum, void* workspace, size_t workspace_bytes,
                  aod_stream_t stream);

/* Reference-precision (x3) launches of aod_conv2d(_ws) with an X-layout destination, N % 128 == 0, a 1x1 or 3x3 filter (forward at any
 * stride, dgrad at stride 1), segments that are whole multiples of 128 rows (but the last), at least ~192 tiles of 128 x 128 and at least
 * 24 K-steps (of 32 channels x 1 tap) per tile -- the 3x3 convs and the deep reduce / lateral 1x1 convs of the trainable backbone stages and
 * of the neck (mmdet/models/backbones/resnet.py:262-301, necks/fpn.py:151-202), their dgrads, and the dgrad of retina_cls
 * (dense_heads/Lambda_L2.py:52) -- run on a persistent producer / consumer kernel
 * (csrc/conv_x3p.hip: loader waves stream the operand tiles through a four-slot LDS ring, consumer waves only issue MFMAs and finish the
 * tile from registers).  Results are bit-identical to the general kernel (the optional column sums are fp32 atomics in both).  Environment
 * AOD_X3P=0 keeps every launch on the general kernel, AOD_X3P_MIN_TILES / AOD_X3P_MIN_STEPS override the two thresholds, AOD_X3P_GROUPED=1
 * also sends the grouped tower launches of aod_conv2d_grouped there (128 x 256 tiles; level with the 256 x 256 tile); with aod_set_deterministic(1) launches
 * that carry column sums stay on the general kernel (ordered sums).  The class-major stride-2 dgrad of a 3x3 / pad-1 conv on an even map and the
 * in-place 1x1 / stride-2 dgrad (res == dst) run there as lattice launches (AOD_X3P_LATTICE=0: general kernel).  aod_conv_x3p_count(): launches the persistent kernel has taken in this
 * process (tests / bench bookkeeping). */
int64_t aod_conv_x3p_count(void);

/* 1x1 / stride-1 / one-segment launches of aod_conv2d(_ws) without zraw / post_scale / fp32 destination are plain GEMMs over consecutive
 * rows (the conv1 / conv3 of every bottleneck, mmdet/models/backbones/resnet.py:260-290, and their dgrads): a persistent streaming
 * kernel (csrc/pointwise.hip) takes them when its tiles fill the CUs.  mode -1 (default): that heuristic (environment AOD_PW_STREAM=0/1
 * overrides), 0: always the general implicit-GEMM kernel, 1: the streaming kernel whenever the shape allows (K % 64 == 0, N % 64 == 0,
 * N <= 2048).  Process-wide; results of the two kernels are bit-identical (tests/test_gpu_kernels.py).  Returns the previous mode. */
int aod_set_pointwise_mode(int mode);

/* Deterministic mode (the reference's `--deterministic`, tools/train_RetinaNet.py:56-68 -> cudnn.deterministic): on = 1 makes the bias /
 * BN-shift column sums of the dgrad epilogues and of the activation-backward / pad-cast passes -- the only order-dependent reduction of
 * the training path (fp32 atomics otherwise) -- sums of per-workgroup partials in a fixed order (csrc/determinism.hip; one small extra
 * launch per vector).  Two runs from the same state are then bit-identical.  Allocates a 32 MB scratch on the CURRENT device the first time
 * it is switched on there (call it outside a graph capture, once per device).  Process-wide; returns the previous setting. */
int aod_set_deterministic(int on);
int aod_get_deterministic(void);

/* A whole 64-channel ResNet bottleneck, forward only (mmdet/models/backbones/resnet.py:262-301 with eval-mode BN; layer1 is frozen,
 * resnet.py:612-628, so nothing of it is needed by the backward pass either):
 *   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + res)
 * x [B*H*W][Cin] bf16 NHWC rows (Cin % 64 == 0), w1 [64][Cin], w2 [64][3][3][64], w3 [256][64] packed forward weights, s* / b* the
 * folded BN scale / shift vectors (fp32), res and y [B*H*W][256] bf16 (res may be x).  One kernel: the two 64-channel intermediates
 * stay in LDS, x and y cross HBM once (csrc/bottleneck.hip). */
int aod_bottleneck64_fwd(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                         const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* res, void* y,
                         aod_stream_t stream);
/* ... for the stage's FIRST block (Cin = 64; resnet.py:291-292): the residual is bn_d(conv_d_1x1(x)), computed inside the launch from the packed
 * [256][64] filter of the downsample conv and its folded BN -- the bits a separate aod_conv2d launch would have stored and this one read back. */
int aod_bottleneck64_ds_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                            const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd,
                            const float* sd, const float* bd, void* y, aod_stream_t stream);
/* The same for an IDENTITY bottleneck of the 128-plane stage (layer2: 512 -> 128 -> 128 -> 512, residual = x): the conv2 / conv3 filters are
 * streamed through LDS rings (csrc/bottleneck_wide.hip).  t1 / t2 (optional, [B*H*W][128] bf16): the block's two intermediates for the
 * pixels of the image -- with them the launch is also the forward of a TRAINING step (the backward pass reads them); NULL: inference. */
int aod_bottleneck128_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                          const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1, void* t2,
                          aod_stream_t stream);
/* ... and of the 256-plane stage (layer3: 1024 -> 256 -> 256 -> 1024; t1 / t2 [B*H*W][256]): 4 x 16 pixel tiles, the eight waves split the
 * output channels, all three filters streamed in 32-KB slices. */
int aod_bottleneck256_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                          const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1, void* t2,
                          aod_stream_t stream);
/* The DGRAD chain of the same identity blocks (planes = 128 / 256) in one launch -- the three dgrad launches of conv3, conv2, conv1 with
 * their fused epilogues (aod_conv2d: mask / res / colsum) back to back, the two intermediates staying in LDS:
 *   gt2 = [act_t2 > 0] * conv3_T(g)        gt1 = [act_t1 > 0] * conv2_T(gt2)        gx = [act_x > 0] * (conv1_T(gt1) + g)
 * g [B*H*W][4*planes]: the finished gradient w.r.t. the block's pre-ReLU output (it is also the skip branch's gradient); wd3 [planes][4*planes],
 * wd2 [planes][3][3][planes], wd1 [4*planes][planes]: the packed dgrad filters (BN scale folded in, aod_pack_weights_dgrad); act_*: the
 * activations the forward pass saved; colsum_*: fp32 [planes] [planes] [4*planes], += column sums of the three results (the BN-shift
 * gradients of conv2, conv1 and of the previous block's conv3).  Gradients equal those of the three launches bit for bit. */
int aod_bottleneck_bwd(int planes, const void* g, int B, int H, int W, const void* wd3, const void* wd2, const void* wd1, const void* act_t2,
                       const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2, float* colsum_t1,
                       float* colsum_x, aod_stream_t stream);

/* The 256-plane block with FRAGMENT-MAJOR filter images: the eight waves of a workgroup split the output channels, so a filter fragment is
 * needed by exactly one wave, in the MFMA A-operand layout a 16-B load per lane delivers -- the filters stream global -> registers (a
 * prefetch ring per wave) and never touch the LDS; the conv2 / conv3 K loops run without a barrier.  aod_frag_pack re-orders row-major
 * packed filters ([rows][K] bf16: aod_pack_weight_fwd / _dgrad images; rows % 256 == 0, K % 64 == 0) into that load order for any number
 * of filters in one launch: items_dev = device array of aod_frag_pack_item_bytes()-byte records { const void* src; void* dst; int32 rows,
 * K, blk0, pad } with blk0 = first block of the item, an item owning ceil(rows * K / 2048) blocks.  Same results as the plain forms, bit
 * for bit. */
int aod_frag_pack_item_bytes(void);
int aod_frag_pack(const void* items_dev, int nitems, int total_blocks, aod_stream_t stream);
int aod_bottleneck256f_fwd(const void* x, int B, int H, int W, const void* w1f, const float* s1, const float* b1, const void* w2f,
                           const float* s2, const float* b2, const void* w3f, const float* s3, const float* b3, void* y, void* t1, void* t2,
                           aod_stream_t stream);
int aod_bottleneck256f_bwd(const void* g, int B, int H, int W, const void* wd3f, const void* wd2f, const void* wd1f, const void* act_t2,
                           const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2, float* colsum_t1,
                           float* colsum_x, aod_stream_t stream);

/* Grouped launch: `ngroups` (<= 4) convolutions with IDENTICAL descriptor (geometry, C, N, filter) but their own operands share one
 * grid -- the cls / reg / evidence towers at one depth (Lambda_L2.py:85-103: three independent 4-conv stacks over the same pyramid).
 * Alone each tower conv leaves a third of its last round of workgroups idle; together their tiles fill whole rounds (3 x 341 tiles of
 * 256 x 256 = 3.996 rounds of the 256 CUs).  bf16 destinations; pre_shift / mask / colsum arrays (or their entries) may be NULL; a
 * dgrad descriptor (transposed = 1, stride 1) is accepted.  Results are bit-identical to `ngroups` aod_conv2d calls. */
int aod_conv2d_grouped(const aod_conv_desc_t* desc, int ngroups, const void* const* src, const void* const* w_packed, void* const* dst,
                       const float* const* pre_shift, const void* const* mask, float* const* colsum, aod_stream_t stream);

/* replaces: the weight-gradient half of autograd's conv backward (cuDNN wgrad) for the same
 * call sites.  dw_f32 is [N][R][S][C] fp32 and is ACCUMULATED into (caller zeroes it);
 * x: forward input [rows, C] bf16; dz: [rows_out, N] bf16. */
int aod_conv2d_wgrad(const aod_conv_desc_t* desc, const void* x, const void* dz, float* dw_f32,
                     const void* row_table, aod_stream_t stream);
/* Deterministic form of the same operation (no float atomics): the kernel splits the pixel axis over aod_conv2d_wgrad_splits(desc)
 * workgroup groups; split s STORES its partial result into slabs + s * slab_stride ([N][R][S][C] fp32 each, need not be initialised;
 * slab_stride >= N*R*S*C elements) and aod_unpack_wgrad_slabs adds the slabs in split order.  Plain stores run at ~6 TB/s where float
 * atomics run at ~1.3 TB/s chip-wide, and the weight gradient becomes bit-reproducible run to run (1-GPU vs N-GPU traces comparable). */
int aod_conv2d_wgrad_splits(const aod_conv_desc_t* desc);
int aod_conv2d_wgrad_slabs(const aod_conv_desc_t* desc, const void* x, const void* dz, float* slabs, int nslabs, int64_t slab_stride,
                           const void* row_table, aod_stream_t stream);
/* Grouped form: the weight gradients of n (<= 4) convolutions of ANY geometries in one grid -- e.g. the three convs of a bottleneck once
 * its dgrad chain has produced all three gradients.  Alone a backbone layer needs 100+ pixel splits of its few tiles to fill the chip and
 * leaves that many partial slabs; together each needs a fraction of them.  aod_conv2d_wgrad_group_plan: splits_out[i] = slabs member i
 * needs; returns 1 (nothing written) when the members do not share a tile form -- launch them one by one then.  All arrays are HOST
 * arrays of n entries.  Results differ from the single launches by the association of the pixel sum only (other split boundaries). */
int aod_conv2d_wgrad_group_plan(const aod_conv_desc_t* const* descs, int n, int32_t* splits_out);
int aod_conv2d_wgrad_grouped(const aod_conv_desc_t* const* descs, int n, const void* const* x, const void* const* dz, float* const* slabs,
                             const int32_t* nslabs, const int64_t* slab_stride, const void* const* row_table, aod_stream_t stream);
/* Row table of a forward descriptor (32 B per destination pixel: source block origin, top-left tap, extents,
 * dZ row).  Depends only on segment geometry / stride / pad / filter size: build once, reuse for every wgrad
 * launch with that geometry. */
size_t aod_conv_row_table_bytes(const aod_conv_desc_t* desc);
int aod_conv_row_table(const aod_conv_desc_t* desc, void* table, aod_stream_t stream);

/* OIHW fp32 -> [O][R][S][Ipad] bf16 (forward) / [I][R][S][Opad] bf16 (dgrad); pads zero-filled, multiples of 8.
 * dgrad: `scale` (nullable, fp32 [O]) multiplies output channel o -- the eval-BN scale gamma*rsqrt(var+eps) of
 * resnet.py:647-656 folded into the weights, so that dX = conv_T(gm, scale*W) needs no scaled copy of the gradient. */
int aod_pack_weight_fwd(const float* w_oihw, void* w_packed, int O, int I, int R, int S, int Ipad, aod_stream_t stream);
int aod_pack_weight_dgrad(const float* w_oihw, void* w_packed, int O, int I, int R, int S, int Opad, const float* scale,
                          aod_stream_t stream);
/* [Opad][R][S][Ipad] fp32 (wgrad result) -> OIHW fp32 gradient (first O rows / I channels), times scale[o] when given;
 * accumulate != 0 adds; clear_src != 0 zeroes every element it reads, so a persistent accumulator is all-zero again.
 * wdot (nullable, fp32 [O], overwritten) receives <w_oihw[o], dw[o]> = sum_m gm[m,o] * z[m,o]; with bn_s1 (= sum_m gm[m,o]),
 * bn_mean and bn_invstd it receives invstd * (that - mean * s1) instead: the weight gradient of the eval-mode BatchNorm behind
 * the conv (resnet.py:262-301) without keeping the pre-BN activations z. */
int aod_unpack_wgrad(float* dw_orsi, float* grad_oihw, int O, int I, int R, int S, int Ipad, int accumulate, int clear_src,
                     const float* scale, const float* w_oihw, float* wdot, const float* bn_s1, const float* bn_mean,
                     const float* bn_invstd, aod_stream_t stream);

/* Slab form (aod_conv2d_wgrad_slabs): dw_slabs = [nslabs][Opad][R][S][Ipad] partial sums, added in slab order; at most 16 taps (the product uses it for 10 .. 16 taps in the reference-precision mode only).
 * The slabs of an x3 launch (aod_conv_desc_t.x3) are LOGICAL -- [N/2][R][S][C/2] for the physical widths N, C of the descriptor, slab_stride
 * >= N*R*S*C/4: the (head, head) + (head, tail) + (tail, head) bands of an entry are added in the wgrad epilogue -- and unpack like any
 * other: Opad = N/2, Ipad = C/2. */
int aod_unpack_wgrad_slabs(const float* dw_slabs, int nslabs, int64_t slab_stride, float* grad_oihw, int O, int I, int R, int S, int Ipad,
                           int accumulate, const float* scale, const float* w_oihw, float* wdot, const float* bn_s1,
                           const float* bn_mean, const float* bn_invstd, aod_stream_t stream);

/* ... and the unpacks of a grouped launch in one grid (arrays of n <= 4 entries; entries of scale / w_oihw / wdot / bn_* may be NULL). */
int aod_unpack_wgrad_slabs_grouped(int n, const float* const* dw_slabs, const int32_t* nslabs, const int64_t* slab_stride, float* const* g,
                                   const int32_t* O, const int32_t* I, const int32_t* R, const int32_t* S, const int32_t* Ipad,
                                   const int32_t* accumulate, const float* const* scale, const float* const* w_oihw, float* const* wdot,
                                   const float* const* bn_s1, const float* const* bn_mean, const float* const* bn_invstd,
                                   aod_stream_t stream);
/* Batched re-derivation of everything the conv launches read from the parameters, for ALL layers in one launch (after an optimizer
 * step every trainable layer is stale): items_dev = device array of `nitems` records of aod_param_prep_item_bytes() bytes,
 *   { const float* w_oihw, gamma, beta, mean, var;  void* w_fwd_packed, w_dgrad_packed;  float* scale, shift, invstd;
 *     int32 O, I, RS, Ipad, Opad, blk0;  float eps;  int32 flags;  int64 pad[2] }
 * flags bit 0: X3 images for aod_conv_desc_t.x3 launches -- the packed channel axis of both images is in the X-layout (Ipad / Opad =
 * the physical widths 2*ceil32(I) / 2*ceil32(O); head = bf16(v), tail = bf16(v - head)) and an item owns ceil32(O)/32 * ceil32(I)/32 blocks.
 * gamma == NULL: conv without BatchNorm (plain packs); w_dgrad_packed == NULL: no dgrad image.  With BN the dgrad image carries
 * scale[o] = gamma*rsqrt(var+eps) (see aod_pack_weight_dgrad) and scale/shift/invstd receive the folded eval-mode BN vectors
 * (resnet.py:647-656).  blk0 = first block of the item; an item owns ceil(Opad/32)*ceil(Ipad/32) blocks (32 x 32 channel tiles), or
 * ceil(max(O*RS*Ipad, I*RS*Opad, O)/2048) blocks when RS > 9. */
int aod_param_prep(const void* items_dev, int nitems, int total_blocks, aod_stream_t stream);
int aod_param_prep_item_bytes(void);

/* ------------------------------------------------------------------ layout / elementwise
 * NCHW fp32 image -> NHWC bf16 with channels zero-padded to Cpad (stem input). replaces the
 * implicit layout of `img` in SSL_L_single_stage.py:45-49 extract_feat. */
int aod_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int B, int C, int H, int W, int Cpad, aod_stream_t stream);
/* Stem input in space-to-depth form: fp32 [B][C <= 4][H][W] (H, W even) -> bf16 [B][H/2][W/2][16], channel slot (dy * 2 + dx) * C + c =
 * pixel (2Y + dy, 2X + dx), remaining slots zero.  The 7x7 / stride-2 / pad-3 stem conv (mmdet/models/backbones/resnet.py:575-600)
 * over the image equals a 4x4 / stride-1 / pad-2 conv over this tensor with the filter taps regrouped the same way. */
int aod_nchw_f32_to_s2d_bf16(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream);
/* The frozen stem in one launch on the space-to-depth image: y = max_pool_3x3_s2_p1(relu(bn1(conv1(x)))) (resnet.py:630-637; conv1 as the
 * 4x4 / stride-1 / pad-2 filter [64][4][4][16] over x_s2d [B][H2][W2][16]).  y [B][H4][W4][64] bf16 with H4 = (H2 - 1) / 2 + 1; the
 * 64-channel conv output never leaves LDS (csrc/stem.hip). */
int aod_stem_pool_fwd(const void* x_s2d, const void* w_packed, const float* scale, const float* shift, void* y, int B, int H2, int W2,
                      aod_stream_t stream);
/* MaxPool 3x3 s2 p1, NHWC bf16 (resnet.py:610). */
int aod_maxpool3x3s2(const void* src, void* dst, int B, int H, int W, int C, aod_stream_t stream);
/* FPN top-down: dst[b,y,x,c] += src[b,y/2,x/2,c] (nearest 2x, fpn.py:163-172) and its adjoint */
int aod_upsample2x_add(const void* src, void* dst, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
int aod_upsample2x_add_bwd(const void* g_dst, void* g_src_accum, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
/* out = lateral + nearest_upsample(top) without touching the lateral (fpn.py:163-172: `laterals[i-1] += F.interpolate(laterals[i])`
 * whose in-place add autograd turns into a copy), and the adjoint that WRITES g_top (no zero fill of the destination). */
int aod_upsample2x_add_to(const void* top, const void* lateral, void* out, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
int aod_upsample2x_add_bwd_set(const void* g_dst, void* g_src, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
/* Backward of y = act(z*scale+shift [+res]) in eval-mode BN (resnet.py:262-301):
 * gm = g * (a > 0 if relu);  dz = gm * scale[n];  dbeta[n] += sum_m gm;  dgamma[n] += sum_m gm * (z - mean[n]) * invstd[n].
 * gmask_out (optional) receives gm (gradient for the residual branch); z/mean/invstd NULL -> bias-only mode
 * (ConvModule bias+ReLU, Lambda_L2.py:44-51): dz = gm, dbeta = column sum.  dbeta/dgamma are fp32, accumulated. */
int aod_act_bwd(const void* g, const void* a, const void* z, const float* scale, const float* mean, const float* invstd,
                void* dz, void* gmask_out, float* dbeta, float* dgamma, int64_t M, int N, int relu, int g_is_f32,
                aod_stream_t stream);
/* dz[M,Npad] (bf16) = pad(g[M,N] * (relu_out > 0 if given));  colsum[n] += sum_m of the same
 * (gradients of the N = 180/36/9 prediction convs; relu_out_f32 = retina_L's fp32 output, Lambda_L2.py:101) */
int aod_pad_cast_colsum(const void* g, const float* relu_out_f32, void* dz, float* colsum, int64_t M, int N, int Npad,
                        int g_is_f32, aod_stream_t stream);
/* out = relu(a + b) bf16 and backward mask (bottleneck join, resnet.py:292-299) */
int aod_add_relu(const void* a, const void* b, void* out, int64_t n, aod_stream_t stream);

/* 3x3 / stride-1 / pad-1 conv with the input halo tile resident in LDS (csrc/halo_conv.hip) for the layers whose output is narrow next to
 * their input: the prediction convs retina_cls / retina_reg / retina_L over all pyramid levels (mmdet/models/dense_heads/Lambda_L2.py:52-54,
 * 92-103) and their dgrads (desc->transposed = 1 with the aod_pack_weight_dgrad packing: taps mirrored).  Same descriptor as aod_conv2d;
 * v = acc + pre_shift[n]; if (mask) v = mask[m][n] > 0 ? v : 0; if (relu) v = max(v, 0); colsum[n] += sum_m v.  desc->C <= 256, N <= 256.
 * aod_halo_conv3x3_applies() tells whether a descriptor qualifies. */
int aod_halo_conv3x3_applies(const aod_conv_desc_t* desc);
int aod_halo_conv3x3(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst, const float* pre_shift,
                     const void* mask, float* colsum, aod_stream_t stream);

/* ------------------------------------------------------------------ losses (K6-K8)
 * replaces: mmcv.ops.sigmoid_focal_loss fwd/bwd CUDA kernels + softmax/log chain at
 *   mmdet/models/losses/EDL_Softmax_FocalLoss.py:9-27,51-69, L1 at smooth_l1_loss.py:33-45,
 *   reductions at losses/utils.py:28-54, called from Lambda_L2.py:112-121.
 * cls [Nrows, C] fp32 logits, labels int64 (C == background), label_w fp32, bbox_* [Nrows,4] fp32.
 * Outputs: loss_noR[Nrows]; sums[0] += sum(l*w), sums[1] += sum(|p-t|*bw), sums[2] += sum(loss_noR)
 * (deterministic two-stage reduction; `partials` is a caller workspace of aod_loss_partials_len floats). */
size_t aod_loss_partials_len(int64_t nrows);
int aod_edl_focal_l1_fwd(const float* cls, const int64_t* labels, const float* label_w,
                         const float* bbox_pred, const float* bbox_tgt, const float* bbox_w,
                         int64_t nrows, int C, float gamma, float alpha,
                         float* loss_noR, float* sums3, float* partials, aod_stream_t stream);
/* grad_cls = d(sum(l*w))*g_cls + d(sum_rows loss_noR * g_noR[row]);  grad_bbox = sign(p-t)*bw*g_bbox.
 * g_cls, g_bbox: device scalars (already divided by avg_factor); g_noR: device [Nrows] -- or ONE device scalar applied to every row when
 * g_noR_is_scalar != 0 (the gradient of mean(loss_noR), Lambda_L2.py:112-121 + SSL_Lambda.py:136-141, without materialising it) -- or NULL,
 * g_noR_scalar: host scalar used when g_noR is NULL.  Output element (row r, col c) is written at
 * (r / A) * pitch + (r % A) * C + c -- i.e. straight into the prediction conv's [pixels, pitch] dZ
 * buffer (pitch = A*C rounded up to 8, pad columns pre-zeroed by the caller) -- as bf16 or fp32. */
int aod_edl_focal_l1_bwd(const float* cls, const int64_t* labels, const float* label_w,
                         const float* bbox_pred, const float* bbox_tgt, const float* bbox_w,
                         int64_t nrows, int C, float gamma, float alpha,
                         const float* g_cls, const float* g_bbox, const float* g_noR, float g_noR_scalar, int g_noR_is_scalar,
                         void* grad_cls, void* grad_bbox, int out_bf16, int A, int pitch_cls, int pitch_box,
                         aod_stream_t stream);
/* Elementwise form [Nrows, C] of the same loss: what EDL_Softmax_FocalLoss.forward(reduction='none') returns
 * (EDL_Softmax_FocalLoss.py:51-69 -> mmcv.ops.sigmoid_focal_loss(..., 'none'), :17).  grad_out == NULL: out[r][c] = l_c (forward);
 * grad_out = upstream gradient [Nrows, C]: out = gradient w.r.t. the logits (backward through softmax -> logit -> focal term). */
int aod_edl_focal_elem(const float* cls, const int64_t* labels, int64_t nrows, int C, float gamma, float alpha,
                       const float* grad_out, float* out, aod_stream_t stream);
/* MEH loss (Lambda_L2.py:235-241): out_sum[0] += sum(((|lam+1e-9-loss|)*w)^2), w = bbox_w4[i*4];
 * grad = g[0]*2*w^2*(lam+1e-9-loss) written at (i / A) * pitch + i % A.  partials: >= aod_loss_partials_len(n) floats */
int aod_meh_loss_fwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n,
                     float* out_sum, float* partials, aod_stream_t stream);
int aod_meh_loss_bwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n,
                     const float* g, void* grad_lam, int out_bf16, int A, int pitch, aod_stream_t stream);
/* The same four passes over ALL pyramid levels in one launch each (the reference runs loss_single / loss_single_L once per level through
 * multi_apply: L_anchor_head.py:306-314,322-327, Lambda_L2.py:105-121,235-241).  The levels' anchor rows are adjacent row ranges of every
 * operand (level l = rows [sum(level_rows[:l]), + level_rows[l]), what the level-batched prediction convs and the level-major targets of
 * aod_max_iou_assign produce); level_rows: HOST array of nlevels <= 8 row counts.  Blocks never straddle a level and every sum runs in the
 * order of a per-level call: loss_noR, the sums and the gradients are bit-identical to nlevels separate calls.
 *   sums [3][nlevels]  (written, not accumulated): row 0 = sum(l*w), row 1 = sum(|p-t|*bw), row 2 = sum(loss_noR) of each level
 *   num_pos (optional, device int32 [num_images], aod_max_iou_assign's per-image positive counts): the sums are then DIVIDED in place --
 *     rows 0, 1 by num_total_samples = sum_b max(num_pos[b], 1) (L_anchor_head.py:300-303, what loss_single divides by at :266-288), row 2 by
 *     the level's row count (the mean of SSL_Lambda.py:136-141) -- with IEEE divisions; divisors [3][nlevels] and num_total [1] are written
 *   g_sums [3][nlevels]: their gradients (device) -- of the divided sums when the forward's `divisors` are passed (else NULL);
 *   g_noR_rows: optional per-row gradient of loss_noR, then used INSTEAD of row 2
 *   out_sums [nlevels] / g [nlevels]: the MEH sums and their gradients.  partials: >= aod_loss_levels_partials_len floats. */
size_t aod_loss_levels_partials_len(int nlevels, const int64_t* level_rows);
int aod_edl_focal_l1_levels_fwd(const float* cls, const int64_t* labels, const float* label_w,
                                const float* bbox_pred, const float* bbox_tgt, const float* bbox_w,
                                int nlevels, const int64_t* level_rows, int C, float gamma, float alpha,
                                float* loss_noR, float* sums, float* partials, const int32_t* num_pos, int num_images,
                                float* divisors, float* num_total, aod_stream_t stream);
int aod_edl_focal_l1_levels_bwd(const float* cls, const int64_t* labels, const float* label_w,
                                const float* bbox_pred, const float* bbox_tgt, const float* bbox_w,
                                int nlevels, const int64_t* level_rows, int C, float gamma, float alpha,
                                const float* g_sums, const float* divisors, const float* g_noR_rows, void* grad_cls, void* grad_bbox, int out_bf16,
                                int A, int pitch_cls, int pitch_box, aod_stream_t stream);
int aod_meh_loss_levels_fwd(const float* lam, const float* loss_noR, const float* bbox_w4, int nlevels, const int64_t* level_rows,
                            float* out_sums, float* partials, aod_stream_t stream);
int aod_meh_loss_levels_bwd(const float* lam, const float* loss_noR, const float* bbox_w4, int nlevels, const int64_t* level_rows,
                            const float* g, void* grad_lam, int out_bf16, int A, int pitch, aod_stream_t stream);

/* ------------------------------------------------------------------ geometry (K9, K10)
 * replaces: AnchorGenerator.grid_anchors/valid_flags (core/anchor/anchor_generator.py:308-438),
 *   MaxIoUAssigner.assign (core/bbox/assigners/max_iou_assigner.py:60-210) + bbox_overlaps
 *   (iou2d_calculator.py:212-252) + PseudoSampler + bbox2delta (delta_xywh_bbox_coder.py:98-140)
 *   + unmap, as driven by L_anchor_head.py:155-257.  Bit-exact integer outputs.
 * anchors [A,4] fp32; valid [B,A] uint8; gts packed [B, Gmax, 4] fp32 with gt_count[B], gt_labels [B,Gmax] int64.
 * outputs: assigned [B,A] int64, labels [B,A] int64, label_w [B,A] f32, bbox_t [B,A,4], bbox_w [B,A,4],
 * num_pos [B] int32.  ws: workspace of aod_assign_ws_bytes(B, Gmax).  nlev > 0 with level_start_host[nlev+1]
 * (anchor offsets of the pyramid levels, HOST array) writes the outputs level-major [L][B][A_l] so that every
 * level's targets are one contiguous block (images_to_levels, anchor/utils.py:4-17, without a copy). */
size_t aod_assign_ws_bytes(int B, int Gmax);
int aod_max_iou_assign(const float* anchors, const uint8_t* valid, int64_t A, int B,
                       const float* gts, const int32_t* gt_count, const int64_t* gt_labels, int Gmax,
                       float pos_thr, float neg_thr, float min_pos_iou, int gt_max_assign_all, int num_classes,
                       const float* means4, const float* stds4,
                       int64_t* assigned, int64_t* labels, float* label_w, float* bbox_t, float* bbox_w,
                       int32_t* num_pos, void* ws, int nlev, const int64_t* level_start_host, aod_stream_t stream);

/* ------------------------------------------------------------------ scoring (K11-K13)
 * replaces Lambda_L2.py:264-304: alphas = softmax; scores = alphas / (sum(alphas) + 1e-20 + 1e-9); row max; level gate
 * (any anchor of the level with max alpha > fg_thr, Lambda_L2.py:497-502).  cls [B, rows_per_img, C] fp32 (one level);
 * rowmax [B, rows_per_img]; any_fg [B] int32 (OR-ed into; caller zeroes).  has_bg = 1 (SSD, My_L_ssd_head.py:331-345): plain softmax
 * over C logits whose last column is background, maxima over the C-1 foreground columns. */
int aod_softmax_rowmax(const float* cls, int B, int64_t rows_per_img, int C, float fg_thr, float* rowmax, int32_t* any_fg,
                       int has_bg, aod_stream_t stream);
/* per image stable top-k (k <= 1024; descending score, ties -> lower index) of score [B, A] -> idx [B, out_pitch] int32
 * (torch.topk at Lambda_L2.py:290; its tie order is unspecified, pinned here) */
int aod_topk_stable(const float* score, int B, int64_t A, int k, int32_t* idx, int64_t out_pitch, aod_stream_t stream);
/* gather + decode ONE level into the concatenated candidate arrays (Lambda_L2.py:292-308,325-326; delta2bbox
 * delta_xywh_bbox_coder.py:144-262): for candidate j (anchor idx[b][j], or j itself when idx == NULL):
 * boxes [B, n_total, 4] (clipped to img_hw[b] = (H, W), divided by scale4[b] when given), scores [B, n_total, C+1]
 * (normalised softmax + zero background column), lam [B, n_total], cand_anchor [B, n_total] = anchor0 + anchor index,
 * written at candidate offset cand0.  normalize: 1 normalised evidence scores, 0 raw softmax (Entropy_ALL), 2 softmax incl. a
 * background logit (SSD: C logits -> C score columns, no padding column). */
int aod_gather_decode(const float* cls, const float* reg, const float* lam_map, const float* anchors, const int32_t* idx,
                      int B, int64_t A, int k, int C, int64_t idx_pitch, const float* img_hw, const float* scale4,
                      const float* means4, const float* stds4, float wh_ratio_clip, float* boxes, float* scores, float* lam,
                      int32_t* cand_anchor, int64_t n_total, int64_t cand0, int64_t anchor0, int normalize, aod_stream_t stream);
/* The three steps above for ALL L (<= 8) pyramid levels of a batch in two launches (the per-level loop of Lambda_L2.py:264-310 /
 * My_L_ssd_head.py:331-365): one scan of every level's logits, then one workgroup per (image, level) that selects the level's top-k[l]
 * (where k[l] < A[l]) and gathers + decodes its candidates.  cls / reg / lam_map / anchors / A / k: HOST arrays of L entries.
 * rowmax: [sum_l B*A[l]] level-major (level l = a [B, A[l]] block); any_fg [L, B] (zeroed by the caller); idx: level-major [B, k[l]]
 * blocks of the levels with k[l] < A[l] only (may be NULL when there is none); n_total = sum_l k[l].  Results are those of the
 * per-level entry points bit for bit. */
int aod_pre_nms_levels(int L, const float* const* cls, const float* const* reg, const float* const* lam_map, const float* const* anchors,
                       const int64_t* A, const int32_t* k, int B, int C, float fg_thr, int has_bg, int normalize, const float* img_hw,
                       const float* scale4, const float* means4, const float* stds4, float wh_ratio_clip, float* rowmax, int32_t* any_fg,
                       int32_t* idx, float* boxes, float* scores, float* lam, int32_t* cand_anchor, int64_t n_total, aod_stream_t stream);
/* multiclass_nms (core/post_processing/bbox_nms.py:7-93 -> mmcv batched_nms / nms_cpu semantics, both its <10000 and
 * per-class paths reduce to this class-aware greedy scan).  boxes [B,n,4], scores [B,n,C+1]; outputs dets [B,max_num,5],
 * det_labels [B,max_num] int64, keep [B,max_num] int64 (index into the score>thr list; -1 padded), num_det [B] int32. */
size_t aod_nms_ws_bytes(int B, int n, int C);
int aod_multiclass_nms(const float* boxes, const float* scores, int B, int n, int C, float score_thr, float iou_thr,
                       int max_num, float* dets, int64_t* det_labels, int64_t* keep, int32_t* num_det,
                       void* ws, aod_stream_t stream);

/* ------------------------------------------------------------------ HUA (K13-K15)
 * replaces GetObjectIdx + ComputeObjUnc + AggregateObjScaleUnc (Lambda_L2.py:343-349,489-537,597-619;
 * torch._sample_dirichlet): per image, (candidate,object) pairs -> num_samples Dirichlet samples -> epistemic ->
 * (object, level, class) bins -> class/scale/object aggregation -> unc[B].
 * level_start_host[L+1]: candidate offsets of the concatenated levels (HOST array); level_any_fg [L][B] int32;
 * cand_anchor [B,n] global anchor id and image_ids [B] int64 key the counter RNG (partition invariant);
 * agg3_host = (class, scale, object) codes 0 Sum / 1 Avg / 2 Max (HOST array; NULL = Sum/Max/Sum).
 * scale_mode = 1: Entropy_ALL / ComputeScaleUnc + AggregateScaleUnc (Lambda_L2.py:539-569,636-691): every candidate whose max
 * score exceeds fg_thr is a pair of one pseudo object, lambda mean over ALL candidates of the level, scores = raw softmax.
 * dirichlet_cols: C (evidence head; 0 = default) or C+1 (SSD: the background probability is a Dirichlet component too).
 * pair_out (optional, [B, max_pairs, 4] f32: cand, obj, aleatoric, epistemic); pair_count [B] int32 (may exceed
 * max_pairs: then the excess pairs were dropped and the caller must retry with a larger workspace). */
size_t aod_hua_ws_bytes(int B, int max_pairs);
int aod_hua_score(const float* boxes, const float* scores, const float* lam, const int32_t* cand_anchor,
                  const float* dets, const int32_t* num_det, const int32_t* level_start_host, const int32_t* level_any_fg,
                  const int64_t* image_ids, int B, int n, int L, int C, int max_num, float obj_score_thr, float obj_iou_thr,
                  float fg_thr, int num_samples, uint64_t seed, const int32_t* agg3_host, int clsW, int scale_mode,
                  int dirichlet_cols, float* unc, float* pair_out, int max_pairs, int32_t* pair_count, void* ws, aod_stream_t stream);

/* ------------------------------------------------------------------ SSD300-VGG16 variant (BASELINE config 0)
 * generic NHWC bf16 max-pool fwd/bwd (mmcv VGG pools, ceil_mode, + the 3x3 s1 p1 pool5 of backbones/ssd_vgg.py:66-68) */
int aod_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k, int s, int p, aod_stream_t stream);
int aod_maxpool_bwd(const void* x, const void* g, void* gx, int B, int H, int W, int C, int OH, int OW, int k, int s, int p,
                    aod_stream_t stream);
/* L2Norm (necks/ssd_neck.py:105-128): y = w[c] * x / (||x||_2 + eps) per pixel; bwd accumulates gw (fp32) */
int aod_l2norm_fwd(const void* x, const float* w, void* y, int64_t rows, int C, float eps, aod_stream_t stream);
int aod_l2norm_bwd(const void* x, const float* w, const void* g, void* gx, float* gw, int64_t rows, int C, float eps, aod_stream_t stream);
/* SSD loss per image (My_L_ssd_head.py:182-215): ce[b][a] = CE(logits, label) * w (== loss_noR); sums3[b] = (sum_pos ce + sum of the
 * min(ratio * #pos, #neg) largest negative ce, sum smooth_l1 * w, mean ce); sel4[b] = selection record consumed by the backward. */
int aod_ssd_loss_fwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred, const float* bbox_tgt,
                     const float* bbox_w, int B, int A, int C1, int num_classes, int neg_pos_ratio, float beta, float* ce, float* sums3,
                     uint32_t* sel4, aod_stream_t stream);
int aod_ssd_loss_bwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred, const float* bbox_tgt,
                     const float* bbox_w, const float* ce, const uint32_t* sel4, int B, int A, int C1, int num_classes, float beta,
                     const float* g_cls, const float* g_box, const float* g_noR, float* grad_cls, float* grad_box, aod_stream_t stream);

/* ------------------------------------------------------------------ synthetic pool (bench harness; nothing to replace in the reference:
 * its pool is VOC images from disk, tools/train_RetinaNet.py:221-225).  dst [B, elems_per_image] fp32 ~ N(0,1), a pure function of
 * (seed, image_ids[b], element index): Philox4x32-10 keyed like the HUA sampler, so a sharded pool scores the same
 * images for any world size.  elems_per_image % 4 == 0. */
int aod_synth_normal_images(float* dst, int B, int64_t elems_per_image, uint64_t seed, const int64_t* image_ids, aod_stream_t stream);

/* ------------------------------------------------------------------ optimizer (K16)
 * torch.optim.SGD semantics (apis/train_Lambda.py:54,59-61): d = g*grad_scale + wd*p; buf = first ? d : mom*buf + d;
 * p -= lr*buf, over HOST arrays of device pointers (params/grads/momentum buffers, fp32) and element counts.
 * lr_dev (nullable): device fp32 scalar that overrides `lr` -- lets a launch captured in a HIP graph follow the LR schedule. */
int aod_sgd_multi(void* const* params, void* const* grads, void* const* moms, const int64_t* sizes, int ntensors,
                  float lr, const float* lr_dev, float momentum, float weight_decay, int first_step, float grad_scale,
                  aod_stream_t stream);

/* ------------------------------------------------------------------ reference-precision mode (aod_conv_desc_t.x3): row kernels on
 * X-layout tensors (csrc/x3_ops.hip).  The reference computes every one of these in fp32 (README.md:13-25); here a value is the fp32 sum of
 * its bf16 head and tail and is written back as such a pair.  `C` = PHYSICAL width (bf16 columns, multiple of 64) unless stated. */
/* fp32 [M][C logical] <-> X rows [M][2*ceil32(C)] (pad channels zero) */
int aod_x3_split(const float* src, void* dst, int64_t M, int C, aod_stream_t stream);
int aod_x3_merge(const void* src, float* dst, int64_t M, int C, aod_stream_t stream);
/* out = a + b over n bf16 elements of X rows: autograd's gradient accumulation where a tensor feeds several consumers (fpn.py:163-202,
 * Lambda_L2.py:85-94) */
int aod_x3_add(const void* a, const void* b, void* out, int64_t n, aod_stream_t stream);
/* aod_nchw_f32_to_s2d_bf16 with X rows of 64 columns (the 16 slots as one 32-channel band) */
int aod_x3_nchw_f32_to_s2d(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream);
/* aod_maxpool3x3s2 / aod_upsample2x_add_to / aod_upsample2x_add_bwd_set on X rows */
int aod_x3_maxpool3x3s2(const void* src, void* dst, int B, int H, int W, int C, aod_stream_t stream);
int aod_x3_upsample2x_add_to(const void* top, const void* lateral, void* out, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
int aod_x3_upsample2x_add_bwd_set(const void* g_dst, void* g_src, int B, int h, int w, int C, int H, int W, aod_stream_t stream);
/* dz = g * [a > 0] (relu != 0; dz may be NULL), colsum[c] += sum_m of it over the C/2 logical channels (aod_act_bwd's masked-gradient part) */
int aod_x3_act_bwd(const void* g, const void* a, void* dz, float* colsum, int64_t M, int C, int relu, aod_stream_t stream);
/* aod_pad_cast_colsum: fp32 head gradients [M][N] (* [relu_out > 0]) -> X rows of 2*ceil32(N) columns + column sums fp32 [ceil32(N)] */
int aod_x3_pad_cast_colsum(const float* g, const float* relu_out_f32, void* dz, float* colsum, int64_t M, int N, aod_stream_t stream);

/* aod_bottleneck64_fwd on X rows (reference-precision mode; csrc/bottleneck_x3.hip): x [B*H*W][2*Cin], w1 / w2 / w3 = the X filter images of
 * aod_param_prep (flags bit 0), res / y [B*H*W][512]; Cin = LOGICAL input channels (64 or 256). */
int aod_bottleneck64x3_fwd(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                           const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* res, void* y,
                           aod_stream_t stream);

/* ... for the stage's FIRST block (Cin = 64; resnet.py:291-292): the residual is bn_d(conv_d_1x1(x)), computed inside the launch from wd = the X
 * filter image [256][128] of the downsample conv and its folded BN -- the bits a separate aod_conv2d launch would have stored and this one read back. */
int aod_bottleneck64x3_ds_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                              const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd,
                              const float* sd, const float* bd, void* y, aod_stream_t stream);

/* aod_bottleneck128_fwd on X rows (reference-precision mode; csrc/bottleneck128_x3.hip): the IDENTITY bottleneck of the 128-plane stage
 * (mmdet/models/backbones/resnet.py:262-301, layer2: 512 -> 128 -> 128 -> 512) in one launch.  x / y [B*H*W][1024], w1 [128][1024], w2
 * [128][9][256], w3 [512][256] = the X filter images of aod_param_prep (flags bit 0), s / b the folded BN vectors; t1 / t2 [B*H*W][256]
 * (both or neither; NULL: inference / frozen) receive the block's two intermediates -- the launch is then the forward of a training step, the
 * three convs being recorded as autograd nodes around its outputs.  Same bits as the three aod_conv2d launches it replaces.  y must not alias x. */
int aod_bottleneck128x3_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                            const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1, void* t2,
                            aod_stream_t stream);

/* aod_bottleneck_bwd (planes = 128) on X rows: the DGRAD chain of the same block in one launch -- the three x3 dgrad launches of conv3, conv2,
 * conv1 with their fused epilogues (mask / res / colsum) back to back, the two intermediate gradients staying in LDS:
 *   gt2 = [act_t2 > 0] * conv3_T(g)        gt1 = [act_t1 > 0] * conv2_T(gt2)        gx = [act_x > 0] * (conv1_T(gt1) + g)
 * g / act_x / gx [B*H*W][1024], act_t2 / act_t1 / gt2 / gt1 [B*H*W][256] X rows; wd3 [128][1024], wd2 [128][3][3][256], wd1 [512][256]: the X dgrad
 * images of aod_param_prep (BN scale folded in before the split); colsum_*: fp32 [128] [128] [512], += column sums of the three results (atomics:
 * the launch stands aside in the deterministic mode).  Gradients equal those of the three launches bit for bit.  gx must not alias g. */
int aod_bottleneck128x3_bwd(const void* g, int B, int H, int W, const void* wd3, const void* wd2, const void* wd1, const void* act_t2,
                            const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2, float* colsum_t1,
                            float* colsum_x, aod_stream_t stream);

/* The narrow prediction convs in the reference-precision mode (Lambda_L2.py:52-54,100-103: retina_reg 256 -> 36, retina_L 256 -> 9 over the five
 * pyramid levels; csrc/halo_x3.hip): x3 forward 3x3 / stride 1 / pad 1, fp32 destination of at most 48 channels, X-layout source rows and X
 * filter image [N][3][3][C].  A K-step is a whole 32-channel chunk (halo + the nine taps' filter slices resident in LDS together).  Same bits as
 * aod_conv2d with the same descriptor.  aod_halo_conv3x3_x3_applies() tells whether a descriptor qualifies. */
int aod_halo_conv3x3_x3_applies(const aod_conv_desc_t* desc);
int aod_halo_conv3x3_x3(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst, const float* pre_shift,
                        aod_stream_t stream);

/* Calibration only (bench.py `roofline.measured_peaks`; no reference call site): `workgroups` x 4 waves issue `iters` x 16 independent bf16
 * MFMAs each and write {shader cycles, 100-MHz ticks} per workgroup to out_u64_pairs -- the clock the chip holds under matrix load
 * (csrc/probe.hip).  sink: one float the kernel may write (keeps the work alive). */
int aod_mfma_clock_probe(int iters, int workgroups, void* out_u64_pairs, float* sink, aod_stream_t stream);

/* The frozen stem of the reference-precision mode in one launch (csrc/stem_x3.hip; resnet.py:630-637): fp32 NCHW image [B][3][H][W] (even
 * H, W) -> y = max_pool_3x3_s2_p1(relu(bn1(conv1_7x7_s2(img)))) as X rows [B][H4][W4][128], H4 = (H/2 - 1) / 2 + 1.  w_x = the X filter
 * image [64][4][4][64] of the space-to-depth form of conv1 (aod_param_prep, flags bit 0, of the [64][12][4][4] filter); replaces
 * aod_x3_nchw_f32_to_s2d + aod_conv2d + aod_x3_maxpool3x3s2 and agrees with them to fp32 summation order. */
int aod_stem_pool_x3_fwd(const float* img, const void* w_x, const float* scale, const float* shift, void* y, int B, int C, int H, int W,
                         aod_stream_t stream);

/* SSD300-VGG16 (BASELINE config 0) in the reference-precision mode: the image as ONE 32-channel band of X rows (64 columns; the first VGG
 * conv reads it), and aod_maxpool_fwd/bwd, aod_l2norm_fwd/bwd on X rows (C = the X-layout width; L2Norm's w has C/2 entries). */
int aod_x3_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream);
int aod_x3_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k, int s, int p, aod_stream_t stream);
int aod_x3_maxpool_bwd(const void* x, const void* g, void* gx, int B, int H, int W, int C, int OH, int OW, int k, int s, int p, aod_stream_t stream);
int aod_x3_l2norm_fwd(const void* x, const float* w, void* y, int64_t rows, int C, float eps, aod_stream_t stream);
int aod_x3_l2norm_bwd(const void* x, const float* w, const void* g, void* gx, float* gw, int64_t rows, int C, float eps, aod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
