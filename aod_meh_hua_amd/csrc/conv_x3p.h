// Persistent producer / consumer form of the reference-precision (x3) convolution: csrc/conv_x3p.hip.  conv.hip's dispatcher fills X3PArgs
// from its own parameter block and asks aod_conv_x3p_wants() whether the launch qualifies.
#pragma once
#include "common.h"

struct X3PArgs {
  const bf16_t* x;          // source rows, X-layout, p.C physical columns
  const bf16_t* w;          // packed filter [N][R][S][C] (physical columns), forward or dgrad image
  bf16_t* y;                // destination rows, X-layout (2 * ceil32(N) columns)
  const float* pre_scale;   // optional fp32 [N]
  const float* pre_shift;   // optional fp32 [N]
  const bf16_t* res;        // optional residual, destination layout
  const bf16_t* mask;       // optional ReLU mask (the producer's saved output), destination layout
  float* colsum;            // optional fp32 [N]: += column sums of the stored values (fp32 atomics)
  int C, N, K, taps, S, stride, pad, dil, transposed, relu, tapin;
  int nseg, M, tiles_m, tiles_n;
  int rot;                  // debug (AOD_X3P_ROT): tile-dependent start chunk of the K loop / dropped operand loads (timing experiments)
  int lat;                  // destination lattice: 0 dense; 1 CLASS-MAJOR stride-2 dgrad of a 3x3 / pad-1 conv (one segment, even OH / OW): the
                            // tiles walk the four (y & 1, x & 1) classes of the destination pixels, a class's tile multiplies only the 1 / 2 / 2 / 4
                            // taps that reach it; 2 the IN-PLACE 1x1 / stride-2 dgrad (conv.hip `up_w`): a plain GEMM over the dZ pixels whose row
                            // (b, y, x) is stored at destination row b * up_hw + 2y * up_w + 2x
  int up_w, up_hw;
  int ngroups;              // >= 1: grp[] holds the operands (the launcher fills grp[0] from the fields above for a plain launch)
  struct { const bf16_t* x; const bf16_t* w; bf16_t* y; const float* shift; const bf16_t* mask; float* colsum; } grp[4];
  long long x_bytes, w_bytes;
  int segH[8], segW[8], segOH[8], segOW[8], segB[8];
  long long seg_src0[8], seg_dst0[8];
  int seg_mend[8];
};

// 1: this launch may take the persistent kernel (shape, operands, segment alignment, fill); 0: the general kernel keeps it
int aod_conv_x3p_wants(const X3PArgs& a, int deterministic_colsum);
int aod_conv_x3p_launch(const X3PArgs& a, hipStream_t st);
