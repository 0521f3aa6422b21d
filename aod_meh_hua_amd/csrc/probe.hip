// Calibration probe (bench.py `roofline.measured_peaks`): the shader clock the chip HOLDS while every SIMD issues bf16 MFMAs back to back.
// MI355X lowers its clock under matrix load (/opt/skills/guides/MI355X_MICROARCH.md, 'DVFS give-back'): the 2.5 PFLOP/s dense bf16 peak is
// quoted at 2.4 GHz, and an MFMA-bound kernel is priced against what the matrix pipe delivers at the clock it actually runs at.  Each workgroup
// (4 waves = one per SIMD, two workgroups per CU) runs `iters` rounds of 16 independent v_mfma_f32_16x16x32_bf16 on pseudo-random register
// operands (zero operands run ~20 % faster: the guide's rule 25) and stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop.
// Nothing of the product path goes through this kernel.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void mfma_clock_probe_kernel(int iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  const unsigned t = threadIdx.x + blockIdx.x * 256u;
  bf16x8 a[4], b[4];
  unsigned s = t * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u; a[i][j] = (bf16_t)(((int)(s >> 9) & 0xffff) * (1.0f / 32768.f) - 1.0f);
      s = s * 1664525u + 1013904223u; b[i][j] = (bf16_t)(((int)(s >> 9) & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v += acc[i][j][0] + acc[i][j][3];
  if (v == 123.456f) sink[0] = v;                  // (keeps the accumulators live)
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
}  // namespace

extern "C" int aod_mfma_clock_probe(int iters, int workgroups, void* out_u64_pairs, float* sink, aod_stream_t stream) {
  AOD_CHECK_ARG(iters >= 1 && workgroups >= 1 && out_u64_pairs && sink, "mfma_clock_probe: bad arguments");
  hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)out_u64_pairs, sink);
  AOD_LAUNCH_CHECK();
  return 0;
}
