// NHWC bf16 implicit-GEMM convolution for gfx950 (MI355X): forward, dgrad (same kernel,
// transposed tap map) and wgrad.  fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Forward / dgrad   D[m][n] = sum_k A[m][k] * Wp[n][k],  m = (segment, b, oy, ox) pixel of the
//   destination, k = (r, s, c) tap-major, A gathered on the fly from the NHWC source (im2col is
//   never materialised), Wp = packed [N][R][S][C] bf16.  256 threads = 4 waves (2x2); tile
//   BM x BN x 64; operands staged global -> VGPR -> LDS (XOR-swizzled 16-B chunks, conflict-free
//   ds_read_b128), double-buffered, one barrier per K-step; epilogue goes through LDS so that
//   every global store is a full 16-B-per-lane row segment (scale/shift/residual/mask/ReLU fused).
// Wgrad   dW[n][(r,s,c)] += sum_m dZ[m][n] * A[m][(r,s,c)]: both operands are contracted over
//   their ROW index, so the LDS images keep the global row-major form and fragments are read
//   with ds_read_b64_tr_b16 (hardware transpose); split over m across workgroups, fp32 atomics.
#include "common.h"

struct ConvKParams {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* pre_scale;
  const float* pre_shift;
  const bf16_t* res;
  const bf16_t* mask;
  const float* post_scale;
  bf16_t* zraw;
  int C, N, K, R, S, stride, pad, dil, transposed, relu, out_f32;
  int nseg, M;
  int tiles_m, tiles_n;
  int segH[8], segW[8], segOH[8], segOW[8], segB[8];
  long long seg_src0[8], seg_dst0[8];
  int seg_mend[8];
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // bijective remap: blocks that share an XCD (bid % 8) get a contiguous range of tiles
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKParams p) {
  constexpr int BK = 64;
  constexpr int CPR = BK / 8;        // 16-B chunks per tile row
  constexpr int RPP = 256 / CPR;     // tile rows covered per pass of the 256 threads
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int ROWB = BK * 2;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int CP = BN + 4;               // fp32 epilogue pitch
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_swizzle(blockIdx.x, nwg);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kc = t % CPR, rb = t / CPR;
  const int C8 = p.C >> 3;

  // ---- per-row gather state (A operand)
  const bf16_t* rptr[A_IT];
  int ry0[A_IT], rx0[A_IT], rH[A_IT], rW[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + rb + i * RPP;
    rH[i] = 0; rW[i] = 0; ry0[i] = 0; rx0[i] = 0; rptr[i] = p.x;
    if (m < p.M) {
      int sg = 0, mstart = 0;
      while (sg < p.nseg - 1 && m >= p.seg_mend[sg]) { mstart = p.seg_mend[sg]; ++sg; }
      const int ml = m - mstart;
      const int ohw = p.segOH[sg] * p.segOW[sg];
      const int b = ml / ohw, rem = ml - b * ohw;
      const int oy = rem / p.segOW[sg], ox = rem - oy * p.segOW[sg];
      rH[i] = p.segH[sg]; rW[i] = p.segW[sg];
      rptr[i] = p.x + (p.seg_src0[sg] + (long long)b * p.segH[sg] * p.segW[sg]) * p.C;
      if (p.transposed) { ry0[i] = oy + p.pad; rx0[i] = ox + p.pad; }
      else { ry0[i] = oy * p.stride - p.pad; rx0[i] = ox * p.stride - p.pad; }
    }
  }
  // tap state of this thread's chunk column
  int c8 = kc, tr = 0, ts = 0;
  while (c8 >= C8) { c8 -= C8; if (++ts == p.S) { ts = 0; ++tr; } }

  const bf16_t* wptr[B_IT];
  bool wok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + rb + i * RPP;
    wok[i] = n < p.N;
    wptr[i] = p.w + (long long)(wok[i] ? n : 0) * p.K + kc * 8;
  }

  bf16x8 areg[A_IT], breg[B_IT];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto gload = [&](int kt) {
    const bool tapok = tr < p.R;
    const int dy = tr * p.dil, dx = ts * p.dil;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int y, x;
      bool ok = tapok;
      if (p.transposed) {
        const int ty = ry0[i] - dy, tx = rx0[i] - dx;
        if (p.stride == 1) { y = ty; x = tx; }
        else { ok = ok && ((ty | tx) >= 0) && (ty % p.stride == 0) && (tx % p.stride == 0); y = ty / p.stride; x = tx / p.stride; }
      } else { y = ry0[i] + dy; x = rx0[i] + dx; }
      ok = ok && (unsigned)y < (unsigned)rH[i] && (unsigned)x < (unsigned)rW[i];
      areg[i] = ok ? *reinterpret_cast<const bf16x8*>(rptr[i] + ((long long)y * rW[i] + x) * p.C + c8 * 8) : zero8;
    }
    const bool kok = (kt * BK + kc * 8) < p.K;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      breg[i] = (wok[i] && kok) ? *reinterpret_cast<const bf16x8*>(wptr[i] + (long long)kt * BK) : zero8;
    // advance tap state by one K-step
    c8 += CPR;
    while (c8 >= C8) { c8 -= C8; if (++ts == p.S) { ts = 0; ++tr; } }
  };
  auto lds_store = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = rb + i * RPP;
      *reinterpret_cast<bf16x8*>(sa + row * ROWB + ((kc ^ ((row >> 1) & 7)) << 4)) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = rb + i * RPP;
      *reinterpret_cast<bf16x8*>(sb + row * ROWB + ((kc ^ ((row >> 1) & 7)) << 4)) = breg[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lds_store(0);
  __syncthreads();
  const int lr = lane & 15, lq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 16 + lr;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + (((ks * 4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WN + j * 16 + lr;
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + (((ks * 4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lds_store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32, [BM][CP]) -> row-major vector stores
  float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sc[(wm * WM + i * 16 + lq * 4 + r) * CP + wn * WN + j * 16 + lr] = acc[i][j][r];
  __syncthreads();

  constexpr int NCH = BN / 8;              // 8-column chunks per tile row
  constexpr int E_IT = BM * NCH / 256;
  const int ec = t % NCH, er = t / NCH;
#pragma unroll 1
  for (int it = 0; it < E_IT; ++it) {
    const int row = er + it * (256 / NCH);
    const int m = m0 + row;
    const int n = n0 + ec * 8;
    if (m >= p.M || n >= p.N) continue;
    int sg = 0, mstart = 0;
    while (sg < p.nseg - 1 && m >= p.seg_mend[sg]) { mstart = p.seg_mend[sg]; ++sg; }
    const long long drow = p.seg_dst0[sg] + (m - mstart);
    float v[8], raw[8];
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    const bool full = (n + 8 <= p.N);
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = v[j];
    if (full) {
      const long long off = drow * p.N + n;
      if (p.pre_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= p.pre_scale[n + j];
      }
      if (p.pre_shift) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += p.pre_shift[n + j];
      }
      if (p.res) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(p.res + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
      }
      if (p.mask) {
        const bf16x8 mv = *reinterpret_cast<const bf16x8*>(p.mask + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)mv[j] > 0.f) ? v[j] : 0.f;
      }
      if (p.post_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= p.post_scale[n + j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(p.y) + off;
        if ((p.N & 3) == 0) {
          *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = v[j];
        }
      } else {
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + off) = ov;
      }
      if (p.zraw) {
        bf16x8 zv;
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[j] = (bf16_t)raw[j];
        *reinterpret_cast<bf16x8*>(p.zraw + off) = zv;
      }
    } else {
      for (int j = 0; j < 8 && n + j < p.N; ++j) {
        const long long off = drow * p.N + n + j;
        float u = v[j];
        if (p.pre_scale) u *= p.pre_scale[n + j];
        if (p.pre_shift) u += p.pre_shift[n + j];
        if (p.res) u += (float)p.res[off];
        if (p.mask) u = ((float)p.mask[off] > 0.f) ? u : 0.f;
        if (p.post_scale) u *= p.post_scale[n + j];
        if (p.relu) u = fmaxf(u, 0.f);
        if (p.out_f32) reinterpret_cast<float*>(p.y)[off] = u;
        else reinterpret_cast<bf16_t*>(p.y)[off] = (bf16_t)u;
        if (p.zraw) p.zraw[off] = (bf16_t)raw[j];
      }
    }
  }
}

template <int BM, int BN>
static int launch_conv(const ConvKParams& p, hipStream_t st) {
  ConvKParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  const size_t stage = (size_t)(BM + BN) * 128 * 2;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  const size_t lds = stage > epi ? stage : epi;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN>), dim3(q.tiles_m * q.tiles_n), dim3(256), lds, st, q);
  return 0;
}

static int fill_params(const aod_conv_desc_t* d, ConvKParams& p) {
  AOD_CHECK_ARG(d->nseg >= 1 && d->nseg <= 8, "conv: nseg %d out of range", d->nseg);
  AOD_CHECK_ARG(d->C % 8 == 0, "conv: source channels %d must be a multiple of 8", d->C);
  AOD_CHECK_ARG(d->stride >= 1 && d->dil >= 1 && d->R >= 1 && d->S >= 1, "conv: bad geometry");
  p.C = d->C; p.N = d->N; p.R = d->R; p.S = d->S; p.K = d->R * d->S * d->C;
  p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.transposed = d->transposed;
  p.relu = d->relu; p.out_f32 = d->out_f32; p.nseg = d->nseg;
  long long m = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const aod_conv_seg_t& s = d->seg[i];
    if (!d->transposed) {
      const int eh = (s.H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
      const int ew = (s.W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
      AOD_CHECK_ARG(eh == s.OH && ew == s.OW, "conv: segment %d output %dx%d != expected %dx%d", i, s.OH, s.OW, eh, ew);
    } else {
      const int eh = (s.OH + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
      const int ew = (s.OW + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
      AOD_CHECK_ARG(eh == s.H && ew == s.W, "dgrad: segment %d dZ %dx%d != expected %dx%d", i, s.H, s.W, eh, ew);
    }
    p.segB[i] = s.B; p.segH[i] = s.H; p.segW[i] = s.W; p.segOH[i] = s.OH; p.segOW[i] = s.OW;
    p.seg_src0[i] = s.src_row0; p.seg_dst0[i] = s.dst_row0;
    m += (long long)s.B * s.OH * s.OW;
    AOD_CHECK_ARG(m < (1ll << 31), "conv: too many rows");
    p.seg_mend[i] = (int)m;
  }
  for (int i = d->nseg; i < 8; ++i) { p.segB[i] = p.segH[i] = p.segW[i] = p.segOH[i] = p.segOW[i] = 0; p.seg_src0[i] = p.seg_dst0[i] = 0; p.seg_mend[i] = (int)m; }
  p.M = (int)m;
  return 0;
}

extern "C" int aod_conv2d(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst,
                          const float* pre_scale, const float* pre_shift, const void* res, const void* mask,
                          const float* post_scale, void* zraw, aod_stream_t stream) {
  AOD_CHECK_ARG(desc && src && w_packed && dst, "conv: null pointer");
  ConvKParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill_params(desc, p);
  if (rc) return rc;
  AOD_CHECK_ARG(!(desc->out_f32 && zraw), "conv: zraw needs a bf16 destination");
  if (p.M == 0) return 0;
  p.x = (const bf16_t*)src; p.w = (const bf16_t*)w_packed; p.y = dst;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.res = (const bf16_t*)res; p.mask = (const bf16_t*)mask;
  p.post_scale = post_scale; p.zraw = (bf16_t*)zraw;
  hipStream_t st = (hipStream_t)stream;
  const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (p.N > 64 && t128 >= 384) launch_conv<128, 128>(p, st);
  else if (p.N > 64) {
    const long long t64 = (long long)((p.M + 63) / 64) * ((p.N + 127) / 128);
    if (t64 >= 2048 || p.M >= 16384) launch_conv<128, 128>(p, st); else launch_conv<64, 128>(p, st);
  } else {
    if ((p.M + 127) / 128 >= 512) launch_conv<128, 64>(p, st); else launch_conv<64, 64>(p, st);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================
// wgrad
// =====================================================================================
struct WgradParams {
  const bf16_t* x;
  const bf16_t* dz;
  float* dw;
  int C, N, K, R, S, stride, pad, dil;
  int nseg, M;
  int tiles_n, tiles_k, splits, rows_per_split;
  int segH[8], segW[8], segOH[8], segOW[8];
  long long seg_src0[8], seg_dst0[8];
  int seg_mend[8];
};

// byte offset of (row, 16-B chunk) in a 256-B-pitch bf16 image that serves transposed reads
__device__ __forceinline__ int tr_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int BKM = 32;                 // contraction rows (pixels) per step
  constexpr int IMG = BKM * 256;          // one [32][128] bf16 image
  constexpr int STAGE = 2 * IMG;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int split = bid % p.splits; bid /= p.splits;
  const int tile_k = bid % p.tiles_k, tile_n = bid / p.tiles_k;
  const int n0 = tile_n * 128, k0 = tile_k * 128;
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  if (ms >= me) return;

  const int ch = t & 15, rb = t >> 4;   // chunk column, row 0..15 (+16)
  // dZ columns of this thread
  const int zn = n0 + ch * 8;
  const bool zok = zn < p.N;            // N % 8 == 0 is required
  // X column (tap, channel) of this thread: fixed for the whole loop
  const int kk = k0 + ch * 8;
  const bool kok = kk < p.K;
  const int tap = kok ? kk / p.C : 0, c0 = kok ? kk - tap * p.C : 0;
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const int dy = tr * p.dil - p.pad, dx = ts * p.dil - p.pad;

  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 zreg[2], xreg[2];
  auto gload = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbase + rb + i * 16;
      zreg[i] = zero8; xreg[i] = zero8;
      if (m < me) {
        int sg = 0, mstart = 0;
        while (sg < p.nseg - 1 && m >= p.seg_mend[sg]) { mstart = p.seg_mend[sg]; ++sg; }
        const int ml = m - mstart;
        if (zok) zreg[i] = *reinterpret_cast<const bf16x8*>(p.dz + (p.seg_dst0[sg] + ml) * (long long)p.N + zn);
        if (kok) {
          const int ohw = p.segOH[sg] * p.segOW[sg];
          const int b = ml / ohw, rem = ml - b * ohw;
          const int oy = rem / p.segOW[sg], ox = rem - oy * p.segOW[sg];
          const int y = oy * p.stride + dy, x = ox * p.stride + dx;
          if ((unsigned)y < (unsigned)p.segH[sg] && (unsigned)x < (unsigned)p.segW[sg])
            xreg[i] = *reinterpret_cast<const bf16x8*>(
                p.x + (p.seg_src0[sg] + ((long long)b * p.segH[sg] + y) * p.segW[sg] + x) * p.C + c0);
        }
      }
    }
  };
  auto lds_store = [&](int buf) {
    char* sz = smem + buf * STAGE;
    char* sx = sz + IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rb + i * 16;
      *reinterpret_cast<bf16x8*>(sz + tr_off(row, ch)) = zreg[i];
      *reinterpret_cast<bf16x8*>(sx + tr_off(row, ch)) = xreg[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = (me - ms + BKM - 1) / BKM;
  gload(ms);
  lds_store(0);
  __syncthreads();
  // transposed-read lane roles: group g = lane>>4 covers k rows 8g..8g+7; lane 4q+p -> row q, cols 4p..4p+3
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  for (int stp = 0; stp < nsteps; ++stp) {
    const int cur = stp & 1;
    if (stp + 1 < nsteps) gload(ms + (stp + 1) * BKM);
    const char* sz = smem + cur * STAGE;
    const char* sx = sz + IMG;
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = wm * 64 + i * 16 + pp * 4;          // n column of the block this lane addresses
      const int r0 = 8 * g + q;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(sz + tr_off(r0, col >> 3) + (col & 7) * 2));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(sz + tr_off(r0 + 4, col >> 3) + (col & 7) * 2));
      af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wn * 64 + j * 16 + pp * 4;
      const int r0 = 8 * g + q;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(sx + tr_off(r0, col >> 3) + (col & 7) * 2));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(sx + tr_off(r0 + 4, col >> 3) + (col & 7) * 2));
      bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (stp + 1 < nsteps) lds_store(cur ^ 1);
    __syncthreads();
  }
  // accumulate into dW[n][kk] (fp32 atomics; one dword per lane, 16 consecutive columns per row group)
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * 64 + i * 16 + lq * 4 + r;
        const int k = k0 + wn * 64 + j * 16 + lr;
        if (n < p.N && k < p.K) atomicAdd(p.dw + (long long)n * p.K + k, acc[i][j][r]);
      }
}

extern "C" int aod_conv2d_wgrad(const aod_conv_desc_t* d, const void* x, const void* dz, float* dw, aod_stream_t stream) {
  AOD_CHECK_ARG(d && x && dz && dw, "wgrad: null pointer");
  AOD_CHECK_ARG(!d->transposed, "wgrad: descriptor must be the forward descriptor");
  AOD_CHECK_ARG(d->N % 8 == 0, "wgrad: N %d must be a multiple of 8 (pad dZ)", d->N);
  ConvKParams cp;
  memset(&cp, 0, sizeof(cp));
  int rc = fill_params(d, cp);
  if (rc) return rc;
  if (cp.M == 0) return 0;
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.dz = (const bf16_t*)dz; p.dw = dw;
  p.C = cp.C; p.N = cp.N; p.K = cp.K; p.R = cp.R; p.S = cp.S; p.stride = cp.stride; p.pad = cp.pad; p.dil = cp.dil;
  p.nseg = cp.nseg; p.M = cp.M;
  for (int i = 0; i < 8; ++i) {
    p.segH[i] = cp.segH[i]; p.segW[i] = cp.segW[i]; p.segOH[i] = cp.segOH[i]; p.segOW[i] = cp.segOW[i];
    p.seg_src0[i] = cp.seg_src0[i]; p.seg_dst0[i] = cp.seg_dst0[i]; p.seg_mend[i] = cp.seg_mend[i];
  }
  p.tiles_n = (p.N + 127) / 128;
  p.tiles_k = (p.K + 127) / 128;
  const int tiles = p.tiles_n * p.tiles_k;
  int splits = (1024 + tiles - 1) / tiles;               // aim at >= 4 workgroups per CU
  const int max_splits = (p.M + 255) / 256;              // at least 8 steps per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (p.M + splits - 1) / splits;
  rps = (rps + 31) / 32 * 32;
  splits = (p.M + rps - 1) / rps;
  p.splits = splits; p.rows_per_split = rps;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
  AOD_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================
// weight re-packing
// =====================================================================================
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, bf16_t* __restrict__ o, int O, int I, int RS, int Ipad) {
  const long long n = (long long)O * RS * Ipad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % Ipad; const long long r1 = i / Ipad; const int rs = r1 % RS; const int oo = r1 / RS;
    o[i] = (c < I) ? (bf16_t)w[((long long)oo * I + c) * RS + rs] : (bf16_t)0.f;
  }
}
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ o, int O, int I, int RS, int Opad) {
  const long long n = (long long)Opad * RS * I;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int oo = i % Opad; const long long r1 = i / Opad; const int rs = r1 % RS; const int c = r1 / RS;
    o[i] = (oo < O) ? (bf16_t)w[((long long)oo * I + c) * RS + rs] : (bf16_t)0.f;
  }
}
__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ g, int O, int I, int RS, int Ipad, int accumulate) {
  const long long n = (long long)O * I * RS;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int rs = i % RS; const long long r1 = i / RS; const int c = r1 % I; const int oo = r1 / I;
    const float v = dw[((long long)oo * RS + rs) * Ipad + c];
    g[i] = accumulate ? g[i] + v : v;
  }
}
static inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

extern "C" int aod_pack_weight_fwd(const float* w, void* o, int O, int I, int R, int S, int Ipad, aod_stream_t stream) {
  AOD_CHECK_ARG(w && o && Ipad >= I && Ipad % 8 == 0, "pack_weight_fwd: bad args");
  hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(grid_for((long long)O * R * S * Ipad)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)o, O, I, R * S, Ipad);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_pack_weight_dgrad(const float* w, void* o, int O, int I, int R, int S, int Opad, aod_stream_t stream) {
  AOD_CHECK_ARG(w && o && Opad >= O && Opad % 8 == 0, "pack_weight_dgrad: Opad must be a multiple of 8 and >= O");
  hipLaunchKernelGGL(pack_w_dgrad_kernel, dim3(grid_for((long long)Opad * R * S * I)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)o, O, I, R * S, Opad);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_unpack_wgrad(const float* dw, float* g, int O, int I, int R, int S, int Ipad, int accumulate, aod_stream_t stream) {
  AOD_CHECK_ARG(dw && g, "unpack_wgrad: null");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for((long long)O * R * S * I)), dim3(256), 0, (hipStream_t)stream, dw, g, O, I, R * S, Ipad, accumulate);
  AOD_LAUNCH_CHECK();
  return 0;
}
