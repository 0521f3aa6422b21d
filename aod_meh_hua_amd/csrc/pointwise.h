// Pointwise (1x1, stride 1, no padding) convolution as a persistent streaming GEMM -- see pointwise.hip.
#pragma once
#include "common.h"

struct PwArgs {
  const bf16_t* x;          // [M][K] bf16, rows consecutive
  const bf16_t* w;          // [N][K] bf16 (the packed forward or dgrad weight of a 1x1 filter)
  bf16_t* y;                // [M][N] bf16
  const float* pre_scale;   // optional fp32 [N]
  const float* pre_shift;   // optional fp32 [N]
  const bf16_t* res;        // optional bf16 [M][N], added before the mask / ReLU
  const bf16_t* mask;       // optional bf16 [M][N]: value kept where mask > 0
  float* colsum;            // optional fp32 [N]: += column sums of the stored values
  int M, N, K, relu;
};

// true when the streaming kernel takes this GEMM (shape limits and the launch heuristic); aod_pw_gemm launches it
bool aod_pw_wants(const PwArgs& a);
int aod_pw_gemm(const PwArgs& a, hipStream_t st);
