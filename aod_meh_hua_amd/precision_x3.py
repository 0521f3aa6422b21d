"""Reference-precision mode `bf16x3` (AOD_CONV_PREC=bf16x3 or functional.set_precision('bf16x3')).

The reference computes in fp32 end to end (README.md:13-25; resnet.py:262-301).  The bf16 matrix pipe is 16x faster than the fp32 one on
gfx950, so fp32-grade arithmetic is built from bf16 pieces: every fp32 value v -- activation, gradient, packed filter element -- travels
as a bf16 head h = bf16(v) and a bf16 tail l = bf16(v - h) (16 significant bits together) and a product is three MFMAs,

        x * w ~= xh * wh + xl * wh + xh * wl            (the dropped xl * wl term is 2^-16 of the product),

summed in the fp32 accumulator.  Round 3 did this as a measuring instrument: operands split and concatenated by torch, fp32 activations,
torch glue between the layers (100 ms per bench step).  Since round 4 it is a KERNEL path:

  * X-layout rows (hipops.xw): 2 * ceil32(C) bf16 columns per pixel, [h(0..31) | l(0..31) | h(32..63) | ...] -- a 64-column K-step of the
    unmodified LDS-DMA staging is then 32 channels with the heads in k-block 0 and the tails in k-block 1, and the kernel only pairs the
    fragments differently (csrc/conv.hip, template flag X3: three MFMAs per K-step instead of two);
  * the epilogues compute in fp32 as before -- folded BN / bias, residual (= head + tail), ReLU, ReLU mask and column sums of the dgrad
    launches -- and store head and tail of every value; split-K, the class-major stride-2 dgrad, segment batching and the gradient
    junctions (functional.GradAcc / ActSlot) work unchanged;
  * wgrad multiplies X rows by X rows: the 128 x 128 tile of the 4-wave form holds the (zh, zl) x (xh, xl) bands of 64 x 64 entries, the
    tail x tail quarter is skipped at compile time, the slab unpack adds the other three (csrc/conv.hip unpack_row);
  * parameter preparation emits X-layout filter images, with the BN scale folded into the dgrad image BEFORE the split (aod_param_prep
    flags bit 0); the fp32 master weights, the optimizer and the loss / scoring kernels (fp32 head outputs) are the product's own;
  * csrc/x3_ops.hip: max-pool, FPN upsample-add and its adjoint, activation backward, head-gradient cast, gradient fan-in add, the
    space-to-depth image -- each re-forms v = h + l, computes in fp32, writes (h, l).

Not in this mode (the launches fall back to the general kernel): the fused bottleneck / stem-pool / halo / pointwise kernels and the
grouped tower launches.  tests/test_gpu_precision_x3.py holds the golden train step and the scoring pass at 1e-4 in this mode."""
from .functional import fork, set_precision, x3_to_f32  # noqa: F401
from .hipops import x3_add, x3_merge, x3_split, xw  # noqa: F401
