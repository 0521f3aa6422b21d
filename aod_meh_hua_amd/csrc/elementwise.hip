// HBM-bound elementwise / reduction kernels around the conv stack (NHWC bf16, 16-B lanes).
#include "common.h"

static inline int grid_for(long long nvec) {
  long long b = (nvec + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC bf16 (padded C)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int HW, int Cpad) {
  const long long n = (long long)B * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, px = i - b * HW;
    bf16_t* o = dst + i * Cpad;
    if (Cpad == 8) {                 // the image (3 channels padded to 8): one 16-B store per pixel instead of eight 2-B stores
      bf16x8 v;
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = (c < C) ? (bf16_t)src[(b * C + c) * HW + px] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(o) = v;
      continue;
    }
    for (int c = 0; c < Cpad; ++c) o[c] = (c < C) ? (bf16_t)src[(b * C + c) * HW + px] : (bf16_t)0.f;
  }
}
extern "C" int aod_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int B, int C, int H, int W, int Cpad, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && Cpad >= C, "nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, C, H * W, Cpad);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- NCHW fp32 -> space-to-depth NHWC bf16 (stem input)
// [B][C <= 4][H][W] -> [B][H/2][W/2][16]: channel slot (dy * 2 + dx) * C + c holds pixel (2Y + dy, 2X + dx); slots >= 4C are zero.  In
// this layout the 7x7 / stride-2 / pad-3 stem conv is a 4x4 / stride-1 / pad-2 conv (tap r = 2R + dy - 1, s = 2S + dx - 1): a filter row
// of an output pixel is 4 x 32 B = one contiguous 128-B line (the 8-channel NHWC form gathers 49 scattered 16-B pieces, every other
// pixel of a row) and K shrinks from 49 x 8 to 16 x 16.  One thread per destination pixel: six 8-B loads, two 16-B stores.
__global__ void nchw_to_s2d_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long long n = (long long)B * H2 * W2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W2); long long r = i / W2;
    const int Y = (int)(r % H2); const long long b = r / H2;
    bf16_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (bf16_t)0.f;
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const float2 p = *reinterpret_cast<const float2*>(src + ((b * C + c) * H + 2 * Y + dy) * (long long)W + 2 * X);
        v[(dy * 2 + 0) * C + c] = (bf16_t)p.x;
        v[(dy * 2 + 1) * C + c] = (bf16_t)p.y;
      }
    bf16x8 lo, hi;
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo[k] = v[k]; hi[k] = v[8 + k]; }
    *reinterpret_cast<bf16x8*>(dst + i * 16) = lo;
    *reinterpret_cast<bf16x8*>(dst + i * 16 + 8) = hi;
  }
}
extern "C" int aod_nchw_f32_to_s2d_bf16(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C >= 1 && C <= 4 && H % 2 == 0 && W % 2 == 0, "nchw_to_s2d: C <= 4 and even H, W required (C %d, %d x %d)", C, H, W);
  hipLaunchKernelGGL(nchw_to_s2d_kernel, dim3(grid_for((long long)B * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, C, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- maxpool 3x3 s2 p1
__global__ void maxpool_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int C8, int OH, int OW) {
  const long long n = (long long)B * OH * OW * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % C8; long long r = i / C8;
    const int ox = r % OW; r /= OW;
    const int oy = r % OH; const int b = r / OH;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int y = oy * 2 - 1 + dy;
      if ((unsigned)y >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int x = ox * 2 - 1 + dx;
        if ((unsigned)x >= (unsigned)W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (((long long)b * H + y) * W + x) * C8 * 8 + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)m[j];
    *reinterpret_cast<bf16x8*>(dst + i * 8) = o;
  }
}
extern "C" int aod_maxpool3x3s2(const void* src, void* dst, int B, int H, int W, int C, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C % 8 == 0, "maxpool: C must be a multiple of 8");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((long long)B * OH * OW * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, B, H, W, C / 8, OH, OW);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- FPN nearest-upsample add (F.interpolate(size=..., 'nearest'))
// src index = floor(dst * h / H) (torch nearest); exact 2x when H == 2h.
// (`lat` = the lateral the upsampled top is added to; it may alias `dst` -- element i is read before it is written by the same thread)
__global__ void upsample_add_kernel(const bf16_t* __restrict__ src, const bf16_t* lat, bf16_t* dst, int B, int h, int w, int C8, int H, int W) {
  const long long n = (long long)B * H * W * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % C8; long long r = i / C8;
    const int x = r % W; r /= W;
    const int y = r % H; const int b = r / H;
    const int sy = min((int)(((long long)y * h) / H), h - 1), sx = min((int)(((long long)x * w) / W), w - 1);
    const bf16x8 s = *reinterpret_cast<const bf16x8*>(src + ((((long long)b * h + sy) * w + sx) * C8 + c) * 8);
    bf16x8 d = *reinterpret_cast<const bf16x8*>(lat + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = (bf16_t)((float)d[j] + (float)s[j]);
    *reinterpret_cast<bf16x8*>(dst + i * 8) = d;
  }
}
extern "C" int aod_upsample2x_add(const void* src, void* dst, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C % 8 == 0, "upsample_add: C must be a multiple of 8");
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((long long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (const bf16_t*)dst, (bf16_t*)dst, B, h, w, C / 8, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}
// out-of-place form: out = lateral + upsample(top) -- the in-place form above needs a copy of the lateral first (autograd must keep the
// lateral conv's output): one pass less over the level
extern "C" int aod_upsample2x_add_to(const void* top, const void* lateral, void* out, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(top && lateral && out && C % 8 == 0, "upsample_add_to: C must be a multiple of 8");
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((long long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)top, (const bf16_t*)lateral, (bf16_t*)out, B, h, w, C / 8, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}
// adjoint: g_src[b,sy,sx,c] (+)= sum over dst pixels mapping to (sy,sx) of g_dst   (ACC: add to what g_src holds; else overwrite it)
template <bool ACC>
__global__ void upsample_add_bwd_kernel(const bf16_t* __restrict__ gd, bf16_t* __restrict__ gs, int B, int h, int w, int C8, int H, int W) {
  const long long n = (long long)B * h * w * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % C8; long long r = i / C8;
    const int sx = r % w; r /= w;
    const int sy = r % h; const int b = r / h;
    // dst rows y with floor(y*h/H) == sy  <=>  y in [ceil(sy*H/h), ceil((sy+1)*H/h))
    const int y0 = (int)(((long long)sy * H + h - 1) / h), y1 = min(H, (int)(((long long)(sy + 1) * H + h - 1) / h));
    const int x0 = (int)(((long long)sx * W + w - 1) / w), x1 = min(W, (int)(((long long)(sx + 1) * W + w - 1) / w));
    float a[8];
    bf16x8 cur;
    if (ACC) cur = *reinterpret_cast<bf16x8*>(gs + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = ACC ? (float)cur[j] : 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(gd + ((((long long)b * H + y) * W + x) * C8 + c) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = (bf16_t)a[j];
    *reinterpret_cast<bf16x8*>(gs + i * 8) = cur;
  }
}
extern "C" int aod_upsample2x_add_bwd(const void* g_dst, void* g_src, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(g_dst && g_src && C % 8 == 0, "upsample_add_bwd: C must be a multiple of 8");
  hipLaunchKernelGGL(upsample_add_bwd_kernel<true>, dim3(grid_for((long long)B * h * w * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)g_dst, (bf16_t*)g_src, B, h, w, C / 8, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}
// ... writing g_src instead of accumulating into it (no zero fill of the destination needed)
extern "C" int aod_upsample2x_add_bwd_set(const void* g_dst, void* g_src, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(g_dst && g_src && C % 8 == 0, "upsample_add_bwd_set: C must be a multiple of 8");
  hipLaunchKernelGGL(upsample_add_bwd_kernel<false>, dim3(grid_for((long long)B * h * w * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)g_dst, (bf16_t*)g_src, B, h, w, C / 8, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- add + relu
__global__ void add_relu_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, long long nvec) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + i * 8), y = *reinterpret_cast<const bf16x8*>(b + i * 8);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)fmaxf((float)x[j] + (float)y[j], 0.f);
    *reinterpret_cast<bf16x8*>(o + i * 8) = r;
  }
}
extern "C" int aod_add_relu(const void* a, const void* b, void* out, int64_t n, aod_stream_t stream) {
  AOD_CHECK_ARG(a && b && out && n % 8 == 0, "add_relu: n must be a multiple of 8");
  hipLaunchKernelGGL(add_relu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long long)(n / 8));
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- activation / BN(eval) backward + column reductions
// Block = 256 threads = 32 column-chunks (8 cols each) x 8 row lanes; each block owns a 256-column
// panel x a strip of rows; column sums are reduced through LDS then one atomicAdd per column per block.
template <bool G_F32>
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ g_, const bf16_t* __restrict__ a, const bf16_t* __restrict__ z,
                                                      const float* __restrict__ scale, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, bf16_t* __restrict__ dz,
                                                      bf16_t* __restrict__ gm_out, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                      long long M, int N, int relu, int rows_per_block, float* __restrict__ cs_ws) {
  __shared__ float sb[8][256 + 8], sg[8][256 + 8];
  const int t = threadIdx.x;
  const int cc = t & 31, rl = t >> 5;
  const int n = blockIdx.x * 256 + cc * 8;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  float sbv[8], sgv[8], sc[8], mu[8], is[8];
  const bool nok = n < N;   // N % 8 == 0
#pragma unroll
  for (int j = 0; j < 8; ++j) { sbv[j] = 0.f; sgv[j] = 0.f; sc[j] = 1.f; mu[j] = 0.f; is[j] = 0.f; }
  if (nok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (scale) sc[j] = scale[n + j];
      if (z) { mu[j] = mean[n + j]; is[j] = invstd[n + j]; }
    }
    for (long long m = r0 + rl; m < r1; m += 8) {
      const long long off = m * N + n;
      float g[8];
      if (G_F32) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>((const float*)g_ + off), g1 = *reinterpret_cast<const f32x4*>((const float*)g_ + off + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = g0[j]; g[4 + j] = g1[j]; }
      } else {
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>((const bf16_t*)g_ + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (float)gv[j];
      }
      if (relu) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = ((float)av[j] > 0.f) ? g[j] : 0.f;
      }
      if (gm_out) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)g[j];
        *reinterpret_cast<bf16x8*>(gm_out + off) = o;
      }
      if (z) {
        const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) sgv[j] += g[j] * ((float)zv[j] - mu[j]) * is[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) { sbv[j] += g[j]; o[j] = (bf16_t)(g[j] * sc[j]); }
      if (dz) *reinterpret_cast<bf16x8*>(dz + off) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { sb[rl][cc * 8 + j] = sbv[j]; sg[rl][cc * 8 + j] = sgv[j]; }
  __syncthreads();
  {
    const int col = t, nn = blockIdx.x * 256 + col;
    if (nn < N) {
      float b = 0.f, gsum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) { b += sb[r][col]; gsum += sg[r][col]; }
      if (cs_ws) {        // deterministic mode: this block's strip is row blockIdx.y of two partial images (dbeta | dgamma), added in order afterwards
        cs_ws[(long long)blockIdx.y * N + nn] = b;
        cs_ws[((long long)gridDim.y + blockIdx.y) * N + nn] = gsum;
      } else {
        if (dbeta) atomicAdd(dbeta + nn, b);
        if (dgamma && z) atomicAdd(dgamma + nn, gsum);
      }
    }
  }
}
extern "C" int aod_act_bwd(const void* g, const void* a, const void* z, const float* scale, const float* mean, const float* invstd,
                           void* dz, void* gmask_out, float* dbeta, float* dgamma, int64_t M, int N, int relu, int g_is_f32,
                           aod_stream_t stream) {
  AOD_CHECK_ARG(g && N % 8 == 0 && M >= 0, "act_bwd: N must be a multiple of 8");
  AOD_CHECK_ARG(!relu || a, "act_bwd: relu needs the forward output");
  AOD_CHECK_ARG(!z || (mean && invstd), "act_bwd: z needs mean/invstd");
  if (M == 0) return 0;
  const int panels = (N + 255) / 256;
  long long want = 2048 / panels;
  if (want < 1) want = 1;
  long long rpb = (M + want - 1) / want;
  if (rpb < 64) rpb = 64;
  rpb = (rpb + 7) / 8 * 8;
  const int gy = (int)((M + rpb - 1) / rpb);
  float* const cs_ws = (dbeta || (dgamma && z)) ? aod_det_scratch((size_t)2 * gy * N) : nullptr;
  if (g_is_f32)
    hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(panels, gy), dim3(256), 0, (hipStream_t)stream, g, (const bf16_t*)a, (const bf16_t*)z, scale, mean,
                       invstd, (bf16_t*)dz, (bf16_t*)gmask_out, dbeta, dgamma, (long long)M, N, relu, (int)rpb, cs_ws);
  else
    hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(panels, gy), dim3(256), 0, (hipStream_t)stream, g, (const bf16_t*)a, (const bf16_t*)z, scale, mean,
                       invstd, (bf16_t*)dz, (bf16_t*)gmask_out, dbeta, dgamma, (long long)M, N, relu, (int)rpb, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) {
    if (dbeta) { const int rc = aod_colsum_finalize(cs_ws, gy, N, N, dbeta, (dgamma && z) ? dgamma : nullptr, (long long)gy * N, (hipStream_t)stream); if (rc) return rc; }
    else if (dgamma && z) return aod_colsum_finalize(cs_ws + (size_t)gy * N, gy, N, N, dgamma, nullptr, 0, (hipStream_t)stream);
  }
  return 0;
}

// ---------------------------------------------------------------- multi-tensor SGD (torch.optim.SGD semantics)
// Pointers travel as kernel arguments (<= 48 tensors per launch): no device-side table, no H2D copy, no sync.
struct SgdChunk {
  float* p[48];
  const float* g[48];
  float* m[48];
  long long n[48];
  int count;
};
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdChunk c, int blocks_per_tensor, float lr_arg, const float* __restrict__ lr_dev,
                                                        float momentum, float wd, int first_step, float grad_scale) {
  const float lr = lr_dev ? *lr_dev : lr_arg;      // device-resident learning rate: a captured launch follows the schedule
  const int ti = blockIdx.x / blocks_per_tensor, bi = blockIdx.x % blocks_per_tensor;
  float* p = c.p[ti];
  const float* g = c.g[ti];
  float* mbuf = c.m[ti];
  const long long n = c.n[ti];
  // 16-B accesses (torch allocations are 512-B aligned; a tensor whose pointers are not 16-B aligned or whose tail is short takes
  // the scalar path): 20 B of traffic per parameter, nothing else
  const bool vec = ((((size_t)p) | ((size_t)g) | ((size_t)mbuf)) & 15) == 0;
  const long long n4 = vec ? n / 4 : 0;
  // four 16-B pieces of each array in flight per thread (one at a time made the big tensors -- 24 iterations per thread -- a chain of memory
  // round trips: the launch with the tower filters took 100 us for 170 MB)
  const long long stride = (long long)blocks_per_tensor * 256;
  for (long long i0 = (long long)bi * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    f32x4 pv[4], gv[4], mv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = i0 + k * stride;
      if (i < n4) {
        pv[k] = reinterpret_cast<const f32x4*>(p)[i]; gv[k] = reinterpret_cast<const f32x4*>(g)[i];
        mv[k] = first_step ? (f32x4){0.f, 0.f, 0.f, 0.f} : reinterpret_cast<const f32x4*>(mbuf)[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = i0 + k * stride;
      if (i < n4) {
        f32x4 po;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = gv[k][u] * grad_scale + wd * pv[k][u];
          const float b = first_step ? d : momentum * mv[k][u] + d;
          mv[k][u] = b;
          po[u] = pv[k][u] - lr * b;
        }
        reinterpret_cast<f32x4*>(mbuf)[i] = mv[k];
        reinterpret_cast<f32x4*>(p)[i] = po;
      }
    }
  }
  for (long long i = n4 * 4 + (long long)bi * 256 + threadIdx.x; i < n; i += (long long)blocks_per_tensor * 256) {
    const float pv = p[i];
    const float d = g[i] * grad_scale + wd * pv;
    const float b = first_step ? d : momentum * mbuf[i] + d;
    mbuf[i] = b;
    p[i] = pv - lr * b;
  }
}
extern "C" int aod_sgd_multi(void* const* params, void* const* grads, void* const* moms, const int64_t* sizes, int ntensors,
                             float lr, const float* lr_dev, float momentum, float weight_decay, int first_step, float grad_scale,
                             aod_stream_t stream) {
  AOD_CHECK_ARG(params && grads && moms && sizes && ntensors >= 0, "sgd_multi: bad args");
  for (int s0 = 0; s0 < ntensors; s0 += 48) {
    SgdChunk c;
    c.count = ntensors - s0 < 48 ? ntensors - s0 : 48;
    long long mx = 0;
    for (int i = 0; i < c.count; ++i) {
      c.p[i] = (float*)params[s0 + i]; c.g[i] = (const float*)grads[s0 + i]; c.m[i] = (float*)moms[s0 + i]; c.n[i] = sizes[s0 + i];
      AOD_CHECK_ARG(c.p[i] && c.g[i] && c.m[i], "sgd_multi: null tensor pointer");
      if (c.n[i] > mx) mx = c.n[i];
    }
    int bpt = (int)((mx + 256 * 16 - 1) / (256 * 16));
    if (bpt < 1) bpt = 1;
    if (bpt > 96) bpt = 96;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(c.count * bpt), dim3(256), 0, (hipStream_t)stream, c, bpt, lr, lr_dev, momentum, weight_decay, first_step, grad_scale);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- pad + cast + column sums (prediction-conv gradients, N = 180 / 36 / 9)
// dz[m][c] = g[m][c] * (a[m][c] > 0 if a given -- fused ReLU of retina_L, Lambda_L2.py:101), zero in the pad columns.
// 256 threads = RP row lanes x TC column lanes (TC = Npad rounded up to a power of two, at most 256): narrow heads (N = 36, 9) keep
// every lane busy, and each thread has four independent rows in flight.
template <bool G_F32>
__global__ __launch_bounds__(256) void pad_cast_colsum_kernel(const void* __restrict__ g_, const float* __restrict__ a, bf16_t* __restrict__ dz,
                                                              float* __restrict__ colsum, long long M, int N, int Npad, int rows_per_block, int TC,
                                                              float* __restrict__ cs_ws) {
  __shared__ float red[256];
  const int RP = 256 / TC;
  const int c0 = threadIdx.x % TC, rl = threadIdx.x / TC;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int cb = 0; cb < Npad; cb += TC) {            // (one pass unless Npad > 256)
    const int c = cb + c0;
    float s = 0.f;
    auto one = [&](long long m) {
      float v = 0.f;
      if (c < N) {
        v = G_F32 ? ((const float*)g_)[m * N + c] : (float)((const bf16_t*)g_)[m * N + c];
        if (a && !(a[m * N + c] > 0.f)) v = 0.f;
      }
      return v;
    };
    if (c < Npad) {
      long long m = r0 + rl;
      for (; m + 3ll * RP < r1; m += 4ll * RP) {
        const float v0 = one(m), v1 = one(m + RP), v2 = one(m + 2ll * RP), v3 = one(m + 3ll * RP);
        dz[m * Npad + c] = (bf16_t)v0; dz[(m + RP) * Npad + c] = (bf16_t)v1;
        dz[(m + 2ll * RP) * Npad + c] = (bf16_t)v2; dz[(m + 3ll * RP) * Npad + c] = (bf16_t)v3;
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; m < r1; m += RP) { const float v = one(m); dz[m * Npad + c] = (bf16_t)v; s += v; }
    }
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
      float tsum = 0.f;
      for (int r = 0; r < RP; ++r) tsum += red[r * TC + c0];
      if (cs_ws) cs_ws[(long long)blockIdx.x * N + c] = tsum;
      else atomicAdd(colsum + c, tsum);
    }
  }
}
extern "C" int aod_pad_cast_colsum(const void* g, const float* relu_out_f32, void* dz, float* colsum, int64_t M, int N, int Npad, int g_is_f32,
                                   aod_stream_t stream) {
  if (M == 0) return 0;
  AOD_CHECK_ARG(g && dz && colsum && Npad >= N && Npad % 8 == 0, "pad_cast_colsum: bad args");
  int rpb = (int)((M + 1023) / 1024);
  if (rpb < 16) rpb = 16;
  const int nb = (int)((M + rpb - 1) / rpb);
  int tc = 8;
  while (tc < Npad && tc < 256) tc <<= 1;
  float* const cs_ws = aod_det_scratch((size_t)nb * N);
  if (g_is_f32) hipLaunchKernelGGL((pad_cast_colsum_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, g, relu_out_f32, (bf16_t*)dz, colsum, (long long)M, N, Npad, rpb, tc, cs_ws);
  else hipLaunchKernelGGL((pad_cast_colsum_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, g, relu_out_f32, (bf16_t*)dz, colsum, (long long)M, N, Npad, rpb, tc, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) return aod_colsum_finalize(cs_ws, nb, N, N, colsum, nullptr, 0, (hipStream_t)stream);
  return 0;
}

// ---------------------------------------------------------------- synthetic pool images (bench harness)
// img[b] ~ N(0, 1) elementwise, generated ON the device from Philox4x32-10 keyed by (seed, global image id, element / 4): a pool image is a
// pure function of its id, so any sharding of the pool over ranks / batches scores the same images.  One Philox block -> two Box-Muller
// pairs -> four consecutive elements (16-B store).
__global__ __launch_bounds__(256) void synth_normal_kernel(float* __restrict__ dst, long long per_img4, unsigned k0, unsigned k1,
                                                           const long long* __restrict__ ids) {
  const int b = blockIdx.y;
  const unsigned id_lo = (unsigned)ids[b], id_hi = (unsigned)((unsigned long long)ids[b] >> 32);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_img4; i += (long long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), id_lo, id_hi, k0, k1, r);
    const float ra = __builtin_amdgcn_sqrtf(-2.f * 0.693147180559945309f * __builtin_amdgcn_logf(u01(r[0])));
    const float rb = __builtin_amdgcn_sqrtf(-2.f * 0.693147180559945309f * __builtin_amdgcn_logf(u01(r[2])));
    f32x4 v;
    v[0] = ra * __builtin_amdgcn_cosf(u01(r[1])); v[1] = ra * __builtin_amdgcn_sinf(u01(r[1]));
    v[2] = rb * __builtin_amdgcn_cosf(u01(r[3])); v[3] = rb * __builtin_amdgcn_sinf(u01(r[3]));
    *reinterpret_cast<f32x4*>(dst + ((long long)b * per_img4 + i) * 4) = v;
  }
}

extern "C" int aod_synth_normal_images(float* dst, int B, int64_t elems_per_image, uint64_t seed, const int64_t* image_ids, aod_stream_t stream) {
  if (B == 0) return 0;
  AOD_CHECK_ARG(dst && image_ids && B > 0 && elems_per_image > 0 && elems_per_image % 4 == 0, "synth_normal_images: elems_per_image must be a positive multiple of 4");
  const long long n4 = elems_per_image / 4;
  const int gx = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(synth_normal_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dst, n4, (unsigned)(seed & 0xffffffffull),
                     (unsigned)(seed >> 32), (const long long*)image_ids);
  AOD_LAUNCH_CHECK();
  return 0;
}
