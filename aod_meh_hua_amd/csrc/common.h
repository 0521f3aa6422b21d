// Shared device/host helpers for libaodhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/aod_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define AOD_WAVE 64

extern thread_local char g_aod_err[512];
int aod_set_err(int code, const char* fmt, ...);

#define AOD_CHECK_ARG(cond, ...)                     \
  do {                                               \
    if (!(cond)) return aod_set_err(-1, __VA_ARGS__); \
  } while (0)

#define AOD_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) return aod_set_err(-3, "%s: %s", __func__, hipGetErrorString(e__)); \
  } while (0)

// hipFuncSetAttribute is per DEVICE: a per-instantiation mask of the devices that already have the attribute (a bare `static bool` left a
// second device of the same process launching with the default 64 KB dynamic-LDS limit)
static inline bool aod_first_on_device(unsigned long long* mask) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
  if ((*mask >> d) & 1ull) return false;
  *mask |= 1ull << d;
  return true;
}

// deterministic mode (determinism.hip): per-device scratch for partial column sums (nullptr when the mode is off or the request is too large:
// the caller keeps its atomic path) and the ordered row sum into the destination vector(s)
float* aod_det_scratch(size_t floats);
int aod_colsum_finalize(const float* ws, int nparts, int pitch, int N, float* dst, float* dst2, long long ws2_off, hipStream_t st);

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block rows [nr][C] (contiguous in global memory) -> LDS rows of pitch P, consecutive lanes on consecutive elements; (row, column) advance
// incrementally (an integer division per element cost more than the load), 16-B loads when a row is a whole number of them
template <int LB>
__device__ __forceinline__ void aod_stage_rows(const float* __restrict__ src, float* __restrict__ srow, int nr, int C, int P) {
  const int tot = nr * C;
  if ((C & 3) == 0 && (reinterpret_cast<unsigned long long>(src) & 15ull) == 0) {
    const int C4 = C >> 2, tot4 = tot >> 2;
    const int drow = LB / C4, dcol = LB - drow * C4;
    int row = (int)threadIdx.x / C4, col = (int)threadIdx.x - row * C4;
    for (int i = threadIdx.x; i < tot4; i += LB) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
      float* d = srow + row * P + 4 * col;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
      row += drow; col += dcol;
      if (col >= C4) { col -= C4; ++row; }
    }
    return;
  }
  const int drow = LB / C, dcol = LB - drow * C;
  int row = (int)threadIdx.x / C, col = (int)threadIdx.x - row * C;
  for (int i = threadIdx.x; i < tot; i += LB) {
    srow[row * P + col] = src[i];
    row += drow; col += dcol;
    if (col >= C) { col -= C; ++row; }
  }
}


// Philox4x32-10 (Salmon et al. 2011), the counter-based RNG of the HUA sampler and of the synthetic pool images; restated in numpy in
// oracle/hua.py (known-answer test: tests/test_oracle_golden.py::test_philox_known_answer)
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* r) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
    const unsigned long long p1 = (unsigned long long)c2 * 0xCD9E8D57ull;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 1.0f) * 5.9604644775390625e-08f; }

