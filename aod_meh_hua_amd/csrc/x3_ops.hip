// Row kernels of the REFERENCE-PRECISION mode (aod_conv_desc_t.x3, conv.hip "X3"): the elementwise / reduction passes around the conv stack on
// X-layout tensors.  An fp32 value v travels as head h = bf16(v) and tail l = bf16(v - h); a tensor of C logical channels is a bf16 row of
// 2 * ceil32(C) columns [h(0..31) | l(0..31) | h(32..63) | l(32..63) | ...].  Every kernel here re-forms v = h + l in fp32, computes like its
// bf16 twin in elementwise.hip and writes (h, l) again: nothing in this mode rounds an activation or a gradient to 8 significant bits.
// Unit of work: one OCTET = 8 consecutive logical channels = a 16-B head piece and the 16-B tail piece 64 B behind it.
#include "common.h"

static inline int grid_for(long long nvec) {
  long long b = (nvec + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// physical column (elements) of logical octet q
__device__ __forceinline__ int xoct(int q) { return ((q >> 2) << 6) + ((q & 3) << 3); }

__device__ __forceinline__ void xload(const bf16_t* __restrict__ p, float (&v)[8]) {
  const bf16x8 h = *reinterpret_cast<const bf16x8*>(p), l = *reinterpret_cast<const bf16x8*>(p + 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
}
__device__ __forceinline__ void xstore(bf16_t* __restrict__ p, const float (&v)[8]) {
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) { h[j] = (bf16_t)v[j]; l[j] = (bf16_t)(v[j] - (float)h[j]); }
  *reinterpret_cast<bf16x8*>(p) = h;
  *reinterpret_cast<bf16x8*>(p + 32) = l;
}

// ---------------------------------------------------------------- fp32 rows <-> X-layout rows
__global__ void x3_split_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long M, int C, int Q) {
  const long long n = M * Q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q); const long long m = i / Q;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (q * 8 + j < C) ? src[m * C + q * 8 + j] : 0.f;
    xstore(dst + m * (Q >> 2) * 64 + xoct(q), v);
  }
}
extern "C" int aod_x3_split(const float* src, void* dst, int64_t M, int C, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C >= 1, "x3_split: bad args");
  if (M == 0) return 0;
  const int Q = (C + 31) / 32 * 4;          // octets of the padded row (pad channels are written as zeros)
  hipLaunchKernelGGL(x3_split_kernel, dim3(grid_for(M * Q)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long long)M, C, Q);
  AOD_LAUNCH_CHECK();
  return 0;
}
__global__ void x3_merge_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long M, int C, int Q) {
  const long long n = M * Q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q); const long long m = i / Q;
    float v[8];
    xload(src + m * (Q >> 2) * 64 + xoct(q), v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (q * 8 + j < C) dst[m * C + q * 8 + j] = v[j];
  }
}
extern "C" int aod_x3_merge(const void* src, float* dst, int64_t M, int C, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C >= 1, "x3_merge: bad args");
  if (M == 0) return 0;
  const int Q = (C + 31) / 32 * 4;
  hipLaunchKernelGGL(x3_merge_kernel, dim3(grid_for(M * Q)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long long)M, C, Q);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- out = a + b (gradient fan-in: a tensor that feeds several consumers)
__global__ void x3_add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, long long nq) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nq; i += (long long)gridDim.x * blockDim.x) {
    const long long off = (i >> 2) * 64 + ((i & 3) << 3);
    float x[8], y[8];
    xload(a + off, x); xload(b + off, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    xstore(o + off, x);
  }
}
extern "C" int aod_x3_add(const void* a, const void* b, void* out, int64_t n, aod_stream_t stream) {
  AOD_CHECK_ARG(a && b && out && n % 64 == 0, "x3_add: n (bf16 elements) must be a multiple of 64");
  if (n == 0) return 0;
  hipLaunchKernelGGL(x3_add_kernel, dim3(grid_for(n / 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long long)(n / 16));
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- stem input: NCHW fp32 image -> space-to-depth X rows
// the 16 slots of aod_nchw_f32_to_s2d_bf16 (slot (dy * 2 + dx) * C + c = pixel (2Y + dy, 2X + dx)) as ONE 32-channel band of the X-layout:
// [h(16 slots) 0 x 16 | l(16 slots) 0 x 16] = 64 columns per space-to-depth pixel
__global__ void x3_nchw_to_s2d_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long long n = (long long)B * H2 * W2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W2); long long r = i / W2;
    const int Y = (int)(r % H2); const long long b = r / H2;
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = 0.f;
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const float2 p = *reinterpret_cast<const float2*>(src + ((b * C + c) * H + 2 * Y + dy) * (long long)W + 2 * X);
        v[(dy * 2 + 0) * C + c] = p.x;
        v[(dy * 2 + 1) * C + c] = p.y;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = v[q * 8 + j];
      xstore(dst + i * 64 + q * 8, u);
    }
  }
}
extern "C" int aod_x3_nchw_f32_to_s2d(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C >= 1 && C <= 4 && H % 2 == 0 && W % 2 == 0, "x3_nchw_to_s2d: C <= 4 and even H, W required (C %d, %d x %d)", C, H, W);
  hipLaunchKernelGGL(x3_nchw_to_s2d_kernel, dim3(grid_for((long long)B * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, C, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- maxpool 3x3 s2 p1 (resnet.py:610)
__global__ void x3_maxpool_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int Q, int OH, int OW) {
  const long long n = (long long)B * OH * OW * Q;
  const int CP = (Q >> 2) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % Q; long long r = i / Q;
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
        float v[8];
        xload(src + (((long long)b * H + y) * W + x) * CP + xoct(q), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    xstore(dst + (i / Q) * CP + xoct(q), m);
  }
}
extern "C" int aod_x3_maxpool3x3s2(const void* src, void* dst, int B, int H, int W, int C, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C % 64 == 0, "x3_maxpool: the X-layout width must be a multiple of 64");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, Q = C / 16;
  hipLaunchKernelGGL(x3_maxpool_kernel, dim3(grid_for((long long)B * OH * OW * Q)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst,
                     B, H, W, Q, OH, OW);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- FPN nearest-upsample add (fpn.py:163-172) and its adjoint
__global__ void x3_upsample_add_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ lat, bf16_t* __restrict__ dst, int B, int h, int w,
                                       int Q, int H, int W) {
  const long long n = (long long)B * H * W * Q;
  const int CP = (Q >> 2) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % Q; long long r = i / Q;
    const long long pix = r;
    const int x = r % W; r /= W;
    const int y = r % H; const int b = r / H;
    const int sy = min((int)(((long long)y * h) / H), h - 1), sx = min((int)(((long long)x * w) / W), w - 1);
    float s[8], d[8];
    xload(src + (((long long)b * h + sy) * w + sx) * CP + xoct(q), s);
    xload(lat + pix * CP + xoct(q), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] += s[j];
    xstore(dst + pix * CP + xoct(q), d);
  }
}
extern "C" int aod_x3_upsample2x_add_to(const void* top, const void* lateral, void* out, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(top && lateral && out && C % 64 == 0, "x3_upsample_add_to: the X-layout width must be a multiple of 64");
  hipLaunchKernelGGL(x3_upsample_add_kernel, dim3(grid_for((long long)B * H * W * (C / 16))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)top,
                     (const bf16_t*)lateral, (bf16_t*)out, B, h, w, C / 16, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}
__global__ void x3_upsample_add_bwd_kernel(const bf16_t* __restrict__ gd, bf16_t* __restrict__ gs, int B, int h, int w, int Q, int H, int W) {
  const long long n = (long long)B * h * w * Q;
  const int CP = (Q >> 2) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % Q; long long r = i / Q;
    const long long pix = r;
    const int sx = r % w; r /= w;
    const int sy = r % h; const int b = r / h;
    const int y0 = (int)(((long long)sy * H + h - 1) / h), y1 = min(H, (int)(((long long)(sy + 1) * H + h - 1) / h));
    const int x0 = (int)(((long long)sx * W + w - 1) / w), x1 = min(W, (int)(((long long)(sx + 1) * W + w - 1) / w));
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        float v[8];
        xload(gd + (((long long)b * H + y) * W + x) * CP + xoct(q), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
      }
    xstore(gs + pix * CP + xoct(q), a);
  }
}
extern "C" int aod_x3_upsample2x_add_bwd_set(const void* g_dst, void* g_src, int B, int h, int w, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(g_dst && g_src && C % 64 == 0, "x3_upsample_add_bwd_set: the X-layout width must be a multiple of 64");
  hipLaunchKernelGGL(x3_upsample_add_bwd_kernel, dim3(grid_for((long long)B * h * w * (C / 16))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g_dst,
                     (bf16_t*)g_src, B, h, w, C / 16, H, W);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- activation backward + column sums: dz = g * [a > 0], colsum[c] += sum_m dz
// Block = 32 octets (256 logical columns) x 8 row lanes over a strip of rows; column sums through LDS, one atomic per column and block.
__global__ __launch_bounds__(256) void x3_act_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ a, bf16_t* __restrict__ dz,
                                                         float* __restrict__ colsum, long long M, int Q, int relu, int rows_per_block, float* __restrict__ cs_ws) {
  __shared__ float sb[8][256 + 8];
  const int t = threadIdx.x, cc = t & 31, rl = t >> 5;
  const int q = blockIdx.x * 32 + cc;
  const int CP = (Q >> 2) * 64;
  const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (q < Q) {
    const int col = xoct(q);
    for (long long m = r0 + rl; m < r1; m += 8) {
      float v[8];
      xload(g + m * CP + col, v);
      if (relu) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + m * CP + col);      // (the head decides the sign)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)av[j] > 0.f) ? v[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
      if (dz) xstore(dz + m * CP + col, v);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sb[rl][cc * 8 + j] = s[j];
  __syncthreads();
  const int n = blockIdx.x * 256 + t;
  if (colsum && n < Q * 8) {
    float b = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) b += sb[r][t];
    if (cs_ws) cs_ws[(long long)blockIdx.y * (Q * 8) + n] = b;      // (deterministic mode: row = this block's strip, added in order by the launcher)
    else atomicAdd(colsum + n, b);
  }
}
extern "C" int aod_x3_act_bwd(const void* g, const void* a, void* dz, float* colsum, int64_t M, int C, int relu, aod_stream_t stream) {
  AOD_CHECK_ARG(g && C % 64 == 0 && M >= 0, "x3_act_bwd: the X-layout width must be a multiple of 64");
  AOD_CHECK_ARG(!relu || a, "x3_act_bwd: relu needs the forward output");
  if (M == 0) return 0;
  const int Q = C / 16, panels = (Q + 31) / 32;
  long long want = 2048 / panels;
  if (want < 1) want = 1;
  long long rpb = (M + want - 1) / want;
  if (rpb < 64) rpb = 64;
  rpb = (rpb + 7) / 8 * 8;
  const int gy = (int)((M + rpb - 1) / rpb);
  float* const cs_ws = colsum ? aod_det_scratch((size_t)gy * Q * 8) : nullptr;
  hipLaunchKernelGGL(x3_act_bwd_kernel, dim3(panels, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)a,
                     (bf16_t*)dz, colsum, (long long)M, Q, relu, (int)rpb, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) return aod_colsum_finalize(cs_ws, gy, Q * 8, Q * 8, colsum, nullptr, 0, (hipStream_t)stream);
  return 0;
}

// ---------------------------------------------------------------- fp32 head gradients -> X rows (+ fused ReLU of retina_L) + column sums
// g fp32 [M][N] (N = 180 / 36 / 9) -> dz X rows of 2 * ceil32(N) columns, zero in the pad channels; colsum fp32 [ceil32(N)]
__global__ __launch_bounds__(256) void x3_pad_cast_colsum_kernel(const float* __restrict__ g, const float* __restrict__ a, bf16_t* __restrict__ dz,
                                                                 float* __restrict__ colsum, long long M, int N, int Np, int rows_per_block, int TC,
                                                                 float* __restrict__ cs_ws) {
  __shared__ float red[256];
  const int RP = 256 / TC;
  const int c0 = threadIdx.x % TC, rl = threadIdx.x / TC;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int cb = 0; cb < Np; cb += TC) {
    const int c = cb + c0;
    const int col = ((c >> 5) << 6) + (c & 31);
    float s = 0.f;
    if (c < Np) {
      for (long long m = r0 + rl; m < r1; m += RP) {
        float v = 0.f;
        if (c < N) {
          v = g[m * N + c];
          if (a && !(a[m * N + c] > 0.f)) v = 0.f;
        }
        const bf16_t h = (bf16_t)v;
        dz[m * 2 * Np + col] = h;
        dz[m * 2 * Np + col + 32] = (bf16_t)(v - (float)h);
        s += v;
      }
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
// ... the same with 8 columns per thread (N % 4 == 0: the fp32 rows are 16-B aligned): two 16-B loads, a 16-B head and a 16-B tail store per
// thread and row instead of eight 4-B loads and sixteen 2-B stores (retina_cls, 180 columns at 16 x 512^2: 80 -> 35 us).  Chunk c8 = columns
// 8 c8 .. + 7 of a band of 32; a chunk that straddles N (176 .. 183 of 180) reads its valid half only.
__global__ __launch_bounds__(256) void x3_pad_cast_colsum_v8_kernel(const float* __restrict__ g, const float* __restrict__ a, bf16_t* __restrict__ dz,
                                                                    float* __restrict__ colsum, long long M, int N, int Np, int rows_per_block, int TC8,
                                                                    float* __restrict__ cs_ws) {
  __shared__ float red[256][9];
  const int RP = 256 / TC8;
  const int c8 = threadIdx.x % TC8, rl = threadIdx.x / TC8;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const int c = c8 * 8;
  const int col = ((c >> 5) << 6) + (c & 31);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < Np) {
    const bool lo = c + 4 <= N, hi = c + 8 <= N;          // (N % 4 == 0: a half chunk is all valid or all pad)
    for (long long m = r0 + rl; m < r1; m += RP) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (lo) v0 = *reinterpret_cast<const f32x4*>(g + m * N + c);
      if (hi) v1 = *reinterpret_cast<const f32x4*>(g + m * N + c + 4);
      if (a) {
        if (lo) { const f32x4 a0 = *reinterpret_cast<const f32x4*>(a + m * N + c);
#pragma unroll
                  for (int j = 0; j < 4; ++j) if (!(a0[j] > 0.f)) v0[j] = 0.f; }
        if (hi) { const f32x4 a1 = *reinterpret_cast<const f32x4*>(a + m * N + c + 4);
#pragma unroll
                  for (int j = 0; j < 4; ++j) if (!(a1[j] > 0.f)) v1[j] = 0.f; }
      }
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
      xstore(dz + m * 2 * Np + col, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (threadIdx.x < N) {
    const int cc = threadIdx.x, q8 = cc >> 3, j = cc & 7;
    float tsum = 0.f;
    for (int r = 0; r < RP; ++r) tsum += red[r * TC8 + q8][j];
    if (cs_ws) cs_ws[(long long)blockIdx.x * N + cc] = tsum;
    else atomicAdd(colsum + cc, tsum);
  }
}

extern "C" int aod_x3_pad_cast_colsum(const float* g, const float* relu_out_f32, void* dz, float* colsum, int64_t M, int N, aod_stream_t stream) {
  if (M == 0) return 0;
  AOD_CHECK_ARG(g && dz && colsum && N >= 1, "x3_pad_cast_colsum: bad args");
  const int Np = (N + 31) / 32 * 32;
  if ((N & 3) == 0 && Np <= 256 && (((size_t)g | (size_t)(relu_out_f32 ? relu_out_f32 : g)) & 15) == 0) {
    int tc8 = 4;                                   // chunks per row, a power of two (idle lanes past Np / 8)
    while (tc8 * 8 < Np) tc8 <<= 1;
    int rpb = (int)((M + 1023) / 1024);
    if (rpb < 64) rpb = 64;
    const int nb = (int)((M + rpb - 1) / rpb);
    float* const cs_ws = aod_det_scratch((size_t)nb * N);
    hipLaunchKernelGGL(x3_pad_cast_colsum_v8_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, relu_out_f32, (bf16_t*)dz, colsum, (long long)M, N, Np,
                       rpb, tc8, cs_ws);
    AOD_LAUNCH_CHECK();
    if (cs_ws) return aod_colsum_finalize(cs_ws, nb, N, N, colsum, nullptr, 0, (hipStream_t)stream);
    return 0;
  }
  int rpb = (int)((M + 1023) / 1024);
  if (rpb < 16) rpb = 16;
  int tc = 32;
  while (tc < Np && tc < 256) tc <<= 1;
  const int nb = (int)((M + rpb - 1) / rpb);
  float* const cs_ws = aod_det_scratch((size_t)nb * N);
  hipLaunchKernelGGL(x3_pad_cast_colsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, relu_out_f32, (bf16_t*)dz, colsum,
                     (long long)M, N, Np, rpb, tc, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) return aod_colsum_finalize(cs_ws, nb, N, N, colsum, nullptr, 0, (hipStream_t)stream);
  return 0;
}

// ---------------------------------------------------------------- SSD300-VGG16 in the reference-precision mode (BASELINE config 0)
// NCHW fp32 image -> X rows: C <= 32 logical channels as ONE band [h(C) 0.. | l(C) 0..] of 64 columns (the 3x3 first conv of VGG reads it)
__global__ void x3_nchw_to_nhwc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int HW) {
  const long long n = (long long)B * HW * 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3); const long long pix = i >> 2;
    const long long b = pix / HW, px = pix - b * HW;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = q * 8 + j; v[j] = c < C ? src[(b * C + c) * HW + px] : 0.f; }
    xstore(dst + pix * 64 + q * 8, v);
  }
}
extern "C" int aod_x3_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(src && dst && C >= 1 && C <= 32, "x3_nchw_to_nhwc: at most 32 channels (got %d)", C);
  hipLaunchKernelGGL(x3_nchw_to_nhwc_kernel, dim3(grid_for((long long)B * H * W * 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, C, H * W);
  AOD_LAUNCH_CHECK();
  return 0;
}

// generic max-pool (k x k, stride s, pad p; torch semantics incl. ceil_mode output sizes computed by the caller) on X rows
__global__ void x3_maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int Q, int OH, int OW, int k, int s, int p) {
  const long long n = (long long)B * OH * OW * Q;
  const int CP = (Q >> 2) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % Q; long long r = i / Q;
    const int ox = r % OW; r /= OW;
    const int oy = r % OH; const int b = r / OH;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
      const int yy = oy * s - p + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int xx = ox * s - p + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        float v[8];
        xload(x + (((long long)b * H + yy) * W + xx) * CP + xoct(q), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    xstore(y + (i / Q) * CP + xoct(q), m);
  }
}
// backward as a gather: input pixel (yy, xx) receives g[oy, ox] from every window whose FIRST maximum (scan order) it is
__global__ void x3_maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g, bf16_t* __restrict__ gx, int B, int H, int W, int Q,
                                      int OH, int OW, int k, int s, int p) {
  const long long n = (long long)B * H * W * Q;
  const int CP = (Q >> 2) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % Q; long long r = i / Q;
    const long long pix = r;
    const int xx = r % W; r /= W;
    const int yy = r % H; const int b = r / H;
    float me[8], acc[8];
    xload(x + pix * CP + xoct(q), me);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int oy0 = yy + p - k + 1; oy0 = oy0 <= 0 ? 0 : (oy0 + s - 1) / s;
    int ox0 = xx + p - k + 1; ox0 = ox0 <= 0 ? 0 : (ox0 + s - 1) / s;
    const int oy1 = min((yy + p) / s, OH - 1), ox1 = min((xx + p) / s, OW - 1);
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        bool win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = true;
        for (int dy = 0; dy < k; ++dy) {
          const int y2 = oy * s - p + dy;
          if ((unsigned)y2 >= (unsigned)H) continue;
          for (int dx = 0; dx < k; ++dx) {
            const int x2 = ox * s - p + dx;
            if ((unsigned)x2 >= (unsigned)W || (y2 == yy && x2 == xx)) continue;
            float v[8];
            xload(x + (((long long)b * H + y2) * W + x2) * CP + xoct(q), v);
            const bool before = (y2 < yy) || (y2 == yy && x2 < xx);
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (v[j] > me[j] || (before && v[j] == me[j])) win[j] = false;
          }
        }
        float gv[8];
        xload(g + (((long long)b * OH + oy) * OW + ox) * CP + xoct(q), gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (win[j]) acc[j] += gv[j];
      }
    xstore(gx + pix * CP + xoct(q), acc);
  }
}
extern "C" int aod_x3_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k, int s, int p, aod_stream_t stream) {
  AOD_CHECK_ARG(x && y && C % 64 == 0 && k >= 1 && s >= 1, "x3_maxpool_fwd: bad args");
  hipLaunchKernelGGL(x3_maxpool_fwd_kernel, dim3(grid_for((long long)B * OH * OW * (C / 16))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, B, H, W, C / 16, OH, OW, k, s, p);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_x3_maxpool_bwd(const void* x, const void* g, void* gx, int B, int H, int W, int C, int OH, int OW, int k, int s, int p,
                                  aod_stream_t stream) {
  AOD_CHECK_ARG(x && g && gx && C % 64 == 0, "x3_maxpool_bwd: bad args");
  hipLaunchKernelGGL(x3_maxpool_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 16))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)g, (bf16_t*)gx, B, H, W, C / 16, OH, OW, k, s, p);
  AOD_LAUNCH_CHECK();
  return 0;
}

// L2Norm (ssd_neck.py:105-128): y = w * x / (||x||_2 + eps) per pixel on X rows; one wavefront per pixel, octets strided over lanes
__global__ __launch_bounds__(256) void x3_l2norm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y, long long rows,
                                                            int Q, float eps) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63, CP = (Q >> 2) * 64;
  float ss = 0.f;
  for (int q = lane; q < Q; q += 64) {
    float v[8];
    xload(x + r * CP + xoct(q), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
  }
  ss = wave_sum(ss);
  const float n = sqrtf(ss) + eps;
  for (int q = lane; q < Q; q += 64) {
    float v[8];
    xload(x + r * CP + xoct(q), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[q * 8 + j] * v[j] / n;
    xstore(y + r * CP + xoct(q), v);
  }
}
// (cs_ws: deterministic mode, see l2norm_bwd_kernel in ssd_ops.hip; Q * 8 <= 1024 logical channels there)
__global__ __launch_bounds__(256) void x3_l2norm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ g,
                                                            bf16_t* __restrict__ gx, float* __restrict__ gw, long long rows, int Q, float eps,
                                                            float* __restrict__ cs_ws) {
  __shared__ float sm[4][1024];
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, CP = (Q >> 2) * 64;
  if (cs_ws) {
    const int wv = threadIdx.x >> 6;
    float n = 1.f, kk = 0.f;
    if (r < rows) {
      float ss = 0.f, dot = 0.f;
      for (int q = lane; q < Q; q += 64) {
        float v[8], gg[8];
        xload(x + r * CP + xoct(q), v); xload(g + r * CP + xoct(q), gg);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ss += v[j] * v[j]; dot += w[q * 8 + j] * gg[j] * v[j]; }
      }
      ss = wave_sum(ss); dot = wave_sum(dot);
      const float nrm = sqrtf(ss);
      n = nrm + eps; kk = nrm > 0.f ? dot / (n * n * nrm) : 0.f;
    }
    for (int q = lane; q < Q; q += 64) {
      float c8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c8[j] = 0.f;
      if (r < rows) {
        float v[8], gg[8], o[8];
        xload(x + r * CP + xoct(q), v); xload(g + r * CP + xoct(q), gg);
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = w[q * 8 + j] * gg[j] / n - kk * v[j]; c8[j] = gg[j] * v[j] / n; }
        xstore(gx + r * CP + xoct(q), o);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[wv][q * 8 + j] = c8[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Q * 8; c += 256) cs_ws[(long long)blockIdx.x * (Q * 8) + c] = ((sm[0][c] + sm[1][c]) + sm[2][c]) + sm[3][c];
    return;
  }
  if (r >= rows) return;
  float ss = 0.f, dot = 0.f;
  for (int q = lane; q < Q; q += 64) {
    float v[8], gg[8];
    xload(x + r * CP + xoct(q), v); xload(g + r * CP + xoct(q), gg);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ss += v[j] * v[j]; dot += w[q * 8 + j] * gg[j] * v[j]; }
  }
  ss = wave_sum(ss); dot = wave_sum(dot);
  const float nrm = sqrtf(ss), n = nrm + eps;
  const float kk = nrm > 0.f ? dot / (n * n * nrm) : 0.f;
  for (int q = lane; q < Q; q += 64) {
    float v[8], gg[8], o[8];
    xload(x + r * CP + xoct(q), v); xload(g + r * CP + xoct(q), gg);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = w[q * 8 + j] * gg[j] / n - kk * v[j]; atomicAdd(gw + q * 8 + j, gg[j] * v[j] / n); }
    xstore(gx + r * CP + xoct(q), o);
  }
}
extern "C" int aod_x3_l2norm_fwd(const void* x, const float* w, void* y, int64_t rows, int C, float eps, aod_stream_t stream) {
  if (rows == 0) return 0;
  AOD_CHECK_ARG(x && w && y && C % 64 == 0, "x3_l2norm_fwd: bad args (C = X-layout width; w has C / 2 entries)");
  hipLaunchKernelGGL(x3_l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)y, (long long)rows, C / 16, eps);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_x3_l2norm_bwd(const void* x, const float* w, const void* g, void* gx, float* gw, int64_t rows, int C, float eps, aod_stream_t stream) {
  if (rows == 0) return 0;
  AOD_CHECK_ARG(x && w && g && gx && gw && C % 64 == 0, "x3_l2norm_bwd: bad args");
  const int nb = (int)((rows + 3) / 4), CL = C / 2;
  float* const cs_ws = CL <= 1024 ? aod_det_scratch((size_t)nb * CL) : nullptr;
  hipLaunchKernelGGL(x3_l2norm_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (const bf16_t*)g, (bf16_t*)gx,
                     gw, (long long)rows, C / 16, eps, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) return aod_colsum_finalize(cs_ws, nb, CL, CL, gw, nullptr, 0, (hipStream_t)stream);
  return 0;
}
