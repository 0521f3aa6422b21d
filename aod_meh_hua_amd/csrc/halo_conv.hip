// 3x3 / stride-1 / pad-1 convolution with the INPUT HALO TILE resident in LDS, for the layers whose output is narrow next to their input:
// the three prediction convs of the RetinaNet / MEH head (Lambda_L2.py:52-54,92-103: retina_cls 256 -> 180, retina_reg 256 -> 36,
// retina_L 256 -> 9 over all five pyramid levels) and their dgrads (184 / 40 / 16 -> 256).
//
// The general implicit-GEMM kernel (conv.hip) gathers every K-step's im2col rows from global memory: for these layers it read the input
// 7.8x (nine taps x three 64-column tiles missing the L2: 348 MB per launch for 44.7 MB of input, profiles/r02_pmc_hbm_traffic_per_launch.json)
// and ran the narrow ones at 6-21 % of the matrix rate although they are input-bound.  Here one 8-wave workgroup owns an 8 x 16 pixel tile
// of one (level, image):
//   * its 10 x 18 halo crosses HBM -> LDS ONCE, in 64-channel chunks (two chunk slots: chunk k + 1 streams in while chunk k is consumed);
//   * K-steps run chunk-major, (64-channel chunk, tap): the pixel operand of a step is read straight from the halo chunk at the tap's offset
//     (row (i + r) * 18 + x + s), only the [N][64] filter slice of the step is streamed (three-slot ring, LDS-DMA, counted s_waitcnt vmcnt(N) +
//     one raw s_barrier per step -- a __syncthreads() would drain the ring);
//   * products are computed transposed (filter fragment = MFMA A operand): a lane ends up with 4 (8 with the paired-block row permutation of
//     bottleneck.hip) consecutive output channels of ONE pixel, so fp32 / bf16 results leave in 16-B pieces straight from the accumulators:
//     no fp32 LDS image, no epilogue barrier;  bias, ReLU, the producer's ReLU mask and the bias-gradient column sums are fused like in conv.hip.
// dgrad = the same kernel on the dgrad-packed filter with the taps mirrored (flip).
#include "common.h"

namespace {

struct HaloSeg { int B, H, W, tiles_x, tiles_per_img, tile0; long long src0, dst0; };
struct HaloArgs {
  const bf16_t* x;       // [rows][C] source rows (one or several pyramid segments)
  const bf16_t* w;       // [N][3][3][C] packed filter (forward packing; for dgrad the [Cin][3][3][Opad] packing with C = Opad)
  void* y;               // [rows][N] fp32 or bf16
  const float* shift;    // [N] or null
  const bf16_t* mask;    // [rows][N] bf16 or null: v = mask > 0 ? v : 0   (dgrad: the producer's ReLU)
  float* colsum;         // [N] or null: += column sums of the stored values
  int C, N, relu, flip, nseg, ntiles;
  long long x_bytes, w_bytes, y_rows;
  HaloSeg seg[8];
};

constexpr int TH = 8, TW = 16, PW_ = TW + 2, PPIX = (TH + 2) * PW_;      // 180 halo pixels
constexpr int PROWS = 192;                                                // padded to 24 groups of 8 rows: 3 LDS-DMA pieces per wave
constexpr int PSLOT = PROWS * 128;                                        // one 64-channel chunk of the halo
constexpr unsigned OOB = 0xf0000000u;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }

__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// NB: 16-channel output blocks (N <= 16 NB); WPX x (8 / WPX) waves over (pixel rows, channel blocks)
template <int NB, int WPX, bool OUT_F32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NB >= 8 ? 2 : 4, NB >= 8 ? 2 : 4))) void halo_conv3x3_kernel(const HaloArgs p) {
  constexpr int WCH = 8 / WPX, NBW = NB / WCH, NPW = TH / WPX;
  static_assert(NB % WCH == 0 && TH % WPX == 0, "wave grid");
  constexpr bool PAIR = (NBW % 2) == 0;                 // paired 16-row blocks: a lane holds 8 consecutive channels
  constexpr int WROWS = (16 * NB + 63) / 64 * 64;       // filter rows per stage, padded so that every wave issues the same number of pieces
  constexpr int NW = WROWS / 64;                        // LDS-DMA pieces per wave and filter stage
  constexpr int NP = PROWS / 64;                        // ... and halo chunk (3)
  constexpr int WSLOT = WROWS * 128;
  constexpr int OFF_W = 2 * PSLOT, OFF_VEC = OFF_W + 3 * WSLOT;
  static_assert(NW + NP <= 7 && 2 * NW <= 8, "wait_vm_dyn range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wp = uw / WCH, wc = uw % WCH;
  const int wg = xcd_remap(blockIdx.x, p.ntiles);
  // (constant indices only: a run-time index into the by-value argument struct would move it to scratch memory)
  int sB = 1, sH = 1, sW = 1, stx = 1, stpi = 1, st0 = 0;
  long long ssrc = 0, sdst = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q < p.nseg && wg >= p.seg[q].tile0) {
      sB = p.seg[q].B; sH = p.seg[q].H; sW = p.seg[q].W; stx = p.seg[q].tiles_x; stpi = p.seg[q].tiles_per_img; st0 = p.seg[q].tile0;
      ssrc = p.seg[q].src0; sdst = p.seg[q].dst0;
    }
  (void)sB;
  const int local = wg - st0;
  const int b = local / stpi, rem = local - b * stpi;
  const int ty0 = (rem / stx) * TH, tx0 = (rem % stx) * TW;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  // bias vector -> LDS (an ordinary global load beside LDS-DMA makes hipcc drain the DMA queue where the value is used: do it first)
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if (t < 256) vec[t] = (p.shift && t < p.N) ? p.shift[t] : 0.f;

  // LDS-DMA lane roles: one wave-instruction fills 8 rows x 8 slots; lane -> row (lane >> 3) of its group, slot lane & 7, source chunk
  // slot ^ key(row); a wave serves row groups uw + 8 i, for which the key is the same
  const int drow = lane >> 3;
  const int kcl = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int kcw = PAIR ? (kcl ^ (((uw >> 1) & 1) << 1)) : kcl;
  const int nsub = (p.C + 63) >> 6;
  const int nsteps = nsub * 9;
  unsigned poff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = 8 * (uw + 8 * i) + drow;
    const int hy = row / PW_, hx = row - hy * PW_;
    const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
    const bool ok = row < PPIX && (unsigned)y < (unsigned)sH && (unsigned)x < (unsigned)sW;
    poff[i] = ok ? (unsigned)((ssrc + ((long long)b * sH + y) * sW + x) * (long long)p.C * 2) : OOB;
  }
  unsigned wbase[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int n = 8 * (uw + 8 * i) + drow;
    wbase[i] = n < p.N ? (unsigned)((long long)n * 9 * p.C * 2) : OOB;
  }
  auto issue_patch = [&](int kc, int slot) {
    const int ch = kc * 8 + kcl;
    const bool cok = ch * 8 < p.C;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const unsigned off = (cok && poff[i] != OOB) ? poff[i] + (unsigned)(ch * 16) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(smem + slot * PSLOT + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
    }
  };
  auto issue_w = [&](int s, int slot) {
    const int kc = s / 9, tap = s - kc * 9;
    const int ch = kc * 8 + kcw;
    const bool cok = ch * 8 < p.C;
    const unsigned koff = (unsigned)((tap * p.C + ch * 8) * 2);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const unsigned off = (cok && wbase[i] != OOB) ? wbase[i] + koff : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + OFF_W + slot * WSLOT + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[NPW][NBW];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue_patch(0, 0);
  issue_w(0, 0);
  if (nsteps > 1) issue_w(1, 1);
  int kc = 0, tap = 0;
  for (int s = 0; s < nsteps; ++s) {
    // younger than this step's filter slice: the next slice (issued one step ago) and a halo chunk issued one or two steps ago
    const bool pfly = (tap == 1 || tap == 2) && kc + 1 < nsub;
    wait_vm_dyn((s + 1 < nsteps ? NW : 0) + (pfly ? NP : 0));
    __builtin_amdgcn_s_barrier();                 // every wave's pieces of this step have landed; every wave is done with step s - 1
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nsteps) issue_w(s + 2, (s + 2) % 3);
    if (tap == 0 && kc + 1 < nsub) issue_patch(kc + 1, (kc + 1) & 1);
    const char* ps = smem + (kc & 1) * PSLOT;
    const char* ws = smem + OFF_W + (s % 3) * WSLOT;
    const int r = tap / 3, q = tap - r * 3;
    const int pr = p.flip ? 2 - r : r, pq = p.flip ? 2 - q : q;
    const int nks = (kc * 64 + 32 < p.C) ? 2 : 1;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 wf[NBW], af[NPW];
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int jg = wc * NBW + j;
        const int row = PAIR ? ((jg >> 1) * 32 + (lr >> 2) * 8 + (jg & 1) * 4 + (lr & 3)) : jg * 16 + lr;
        wf[j] = *reinterpret_cast<const bf16x8*>(ws + (PAIR ? wswz(row, ks * 4 + lq) : swz(row, ks * 4 + lq)));
      }
#pragma unroll
      for (int i = 0; i < NPW; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ps + swz((wp * NPW + i + pr) * PW_ + lr + pq, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < NPW; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads are done before it passes the next barrier
    if (++tap == 9) { tap = 0; ++kc; }
  }

  // ---- epilogue straight from the accumulators: a lane holds CW consecutive channels of one pixel per unit
  constexpr int CW = PAIR ? 8 : 4;
  constexpr int NU = PAIR ? NBW / 2 : NBW;
  float csum[NU][CW];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < CW; ++r) csum[u][r] = 0.f;
  const bool nvec = (p.N % CW) == 0 || (OUT_F32 && (p.N & 3) == 0);      // rows keep the store alignment
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int y = ty0 + wp * NPW + i, x = tx0 + lr;
    const bool ok = y < sH && x < sW;
    const long long orow = sdst + ((long long)b * sH + y) * sW + x;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int jg0 = wc * NBW + (PAIR ? 2 * u : u);
      const int ch0 = PAIR ? ((jg0 >> 1) * 32 + lq * 8) : jg0 * 16 + lq * 4;
      float v[CW];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][PAIR ? 2 * u : u][r] + vec[(ch0 + r) & 255];
        if (PAIR) v[4 + r] = acc[i][PAIR ? 2 * u + 1 : u][r] + vec[(ch0 + 4 + r) & 255];
      }
      const bool full = ch0 + CW <= p.N;
      if (p.mask && ok) {
        if (full && (p.N % CW) == 0) {
          if (PAIR) {
            const bf16x8 mv = *reinterpret_cast<const bf16x8*>(p.mask + orow * p.N + ch0);
#pragma unroll
            for (int r = 0; r < CW; ++r) v[r] = ((float)mv[r] > 0.f) ? v[r] : 0.f;
          } else {
            const bf16x4 mv = *reinterpret_cast<const bf16x4*>(p.mask + orow * p.N + ch0);
#pragma unroll
            for (int r = 0; r < CW; ++r) v[r] = ((float)mv[r] > 0.f) ? v[r] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < CW; ++r)
            if (ch0 + r < p.N) v[r] = ((float)p.mask[orow * p.N + ch0 + r] > 0.f) ? v[r] : 0.f;
        }
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < CW; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (ok) {
#pragma unroll
        for (int r = 0; r < CW; ++r) csum[u][r] += v[r];
        if (OUT_F32) {
          float* o = reinterpret_cast<float*>(p.y) + orow * p.N + ch0;
          if (nvec && full) {
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
            if (PAIR) *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v[4 % CW], v[5 % CW], v[6 % CW], v[7 % CW]};
          } else if (nvec && PAIR && ch0 + 4 <= p.N) {                       // (N = 180: the last unit holds 4 real channels)
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
#pragma unroll
            for (int r = 4; r < CW; ++r) if (ch0 + r < p.N) o[r] = v[r];
          } else {
#pragma unroll
            for (int r = 0; r < CW; ++r) if (ch0 + r < p.N) o[r] = v[r];
          }
        } else {
          bf16_t* o = reinterpret_cast<bf16_t*>(p.y) + orow * p.N + ch0;
          if (nvec && full) {
            if (PAIR) {
              bf16x8 ov;
#pragma unroll
              for (int r = 0; r < CW; ++r) ov[r] = (bf16_t)v[r];
              *reinterpret_cast<bf16x8*>(o) = ov;
            } else {
              bf16x4 ov;
#pragma unroll
              for (int r = 0; r < CW; ++r) ov[r] = (bf16_t)v[r];
              *reinterpret_cast<bf16x4*>(o) = ov;
            }
          } else {
#pragma unroll
            for (int r = 0; r < CW; ++r) if (ch0 + r < p.N) o[r] = (bf16_t)v[r];
          }
        }
      }
    }
  }
  if (p.colsum) {
    // column sums of the stored values: 16 pixel lanes -> lane lr == 0, then the WPX pixel-row waves through LDS in wave order, one atomic
    // per channel and workgroup.  The staging area [WPX][256] floats overlays the filter ring: every wave must be out of the K loop first.
    __syncthreads();
    float* const csw = reinterpret_cast<float*>(smem + OFF_W);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int jg0 = wc * NBW + (PAIR ? 2 * u : u);
      const int ch0 = PAIR ? ((jg0 >> 1) * 32 + lq * 8) : jg0 * 16 + lq * 4;
#pragma unroll
      for (int r = 0; r < CW; ++r) {
        float v = csum[u][r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (lr == 0) csw[wp * 256 + ((ch0 + r) & 255)] = v;
      }
    }
    __syncthreads();
    if (t < p.N && t < 256) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPX; ++w) s += csw[w * 256 + t];
      atomicAdd(p.colsum + t, s);
    }
  }
}

template <int NB, int WPX, bool OUT_F32>
int launch_halo(const HaloArgs& a, hipStream_t st) {
  constexpr int WROWS = (16 * NB + 63) / 64 * 64;
  constexpr int LDS = 2 * PSLOT + 3 * WROWS * 128 + 1024;
  static_assert(LDS <= 160 * 1024 && 3 * WROWS * 128 >= WPX * 1024, "LDS map (the column-sum staging overlays the filter ring)");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo_conv3x3_kernel<NB, WPX, OUT_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL((halo_conv3x3_kernel<NB, WPX, OUT_F32>), dim3(a.ntiles), dim3(512), LDS, st, a);
  return 0;
}

}  // namespace

// 1 when aod_halo_conv3x3 handles this descriptor (3x3 / stride 1 / pad 1, <= 256 source channels, <= 256 output channels)
extern "C" int aod_halo_conv3x3_applies(const aod_conv_desc_t* d) {
  if (!d || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->nseg < 1 || d->nseg > 8) return 0;
  if (d->C % 8 != 0 || d->C < 8 || d->C > 256 || d->N < 1 || d->N > 256) return 0;
  for (int i = 0; i < d->nseg; ++i)
    if (d->seg[i].OH != d->seg[i].H || d->seg[i].OW != d->seg[i].W) return 0;
  return 1;
}

extern "C" int aod_halo_conv3x3(const aod_conv_desc_t* d, const void* src, const void* w_packed, void* dst, const float* pre_shift,
                                const void* mask, float* colsum, aod_stream_t stream) {
  AOD_CHECK_ARG(d && src && w_packed && dst, "halo_conv: null pointer");
  AOD_CHECK_ARG(aod_halo_conv3x3_applies(d), "halo_conv: needs a 3x3 / stride-1 / pad-1 conv with C <= 256 (multiple of 8) and N <= 256");
  HaloArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)src; a.w = (const bf16_t*)w_packed; a.y = dst; a.shift = pre_shift; a.mask = (const bf16_t*)mask; a.colsum = colsum;
  a.C = d->C; a.N = d->N; a.relu = d->relu; a.flip = d->transposed ? 1 : 0; a.nseg = d->nseg;
  long long xrows = 0, yrows = 0;
  int tiles = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const aod_conv_seg_t& s = d->seg[i];
    AOD_CHECK_ARG(s.src_row0 >= 0 && s.dst_row0 >= 0 && s.B >= 0, "halo_conv: negative row offset");
    HaloSeg& h = a.seg[i];
    h.B = s.B; h.H = s.H; h.W = s.W; h.src0 = s.src_row0; h.dst0 = s.dst_row0;
    h.tiles_x = (s.W + TW - 1) / TW;
    h.tiles_per_img = h.tiles_x * ((s.H + TH - 1) / TH);
    h.tile0 = tiles;
    tiles += s.B * h.tiles_per_img;
    const long long e = s.src_row0 + (long long)s.B * s.H * s.W, ey = s.dst_row0 + (long long)s.B * s.H * s.W;
    if (e > xrows) xrows = e;
    if (ey > yrows) yrows = ey;
  }
  for (int i = d->nseg; i < 8; ++i) a.seg[i].tile0 = 0x7fffffff;
  if (tiles == 0) return 0;
  a.ntiles = tiles;
  a.x_bytes = xrows * a.C * 2;
  a.w_bytes = (long long)a.N * 9 * a.C * 2;
  a.y_rows = yrows;
  AOD_CHECK_ARG(a.x_bytes < 0xe0000000ll && a.w_bytes < 0xe0000000ll, "halo_conv: operand larger than 3.5 GiB (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (a.N + 15) / 16;
  if (d->out_f32) {
    if (nb <= 1) launch_halo<1, 8, true>(a, st);
    else if (nb <= 4) launch_halo<4, 4, true>(a, st);
    else if (nb <= 12) launch_halo<12, 4, true>(a, st);
    else launch_halo<16, 4, true>(a, st);
  } else {
    if (nb <= 1) launch_halo<1, 8, false>(a, st);
    else if (nb <= 4) launch_halo<4, 4, false>(a, st);
    else if (nb <= 12) launch_halo<12, 4, false>(a, st);
    else launch_halo<16, 4, false>(a, st);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}
