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
constexpr int PROWS = 184;                                                // 23 groups of 8 rows (waves 0-6 move 3 LDS-DMA pieces per chunk, wave 7 two)
constexpr int PSLOT = PROWS * 128;                                        // one 64-channel chunk of the halo
constexpr unsigned OOB = 0xf0000000u;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }

#define AOD_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    AOD_VMCASE(0) AOD_VMCASE(1) AOD_VMCASE(2) AOD_VMCASE(3) AOD_VMCASE(4) AOD_VMCASE(5) AOD_VMCASE(6) AOD_VMCASE(7) AOD_VMCASE(8)
    AOD_VMCASE(9) AOD_VMCASE(10) AOD_VMCASE(11) AOD_VMCASE(12) AOD_VMCASE(13) AOD_VMCASE(14) AOD_VMCASE(15) AOD_VMCASE(16)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef AOD_VMCASE

// NB: 16-channel output blocks per PASS (the kernel makes ceil(N / 16 NB) passes over the K loop; the halo is re-streamed per pass, from L2);
// WPX x (8 / WPX) waves over (pixel rows, channel blocks); R: slots of the filter ring (R - 1 filter slices in flight: a K-step of a narrow
// layer is a few dozen MFMAs, far shorter than an L2 round trip).
template <int NB, int WPX, int R, bool OUT_F32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NB >= 8 ? 2 : 4, NB >= 8 ? 2 : 4))) void halo_conv3x3_kernel(const HaloArgs p) {
  constexpr int WCH = 8 / WPX, NBW = NB / WCH, NPW = TH / WPX;
  static_assert(NB % WCH == 0 && TH % WPX == 0, "wave grid");
  constexpr bool PAIR = (NBW % 2) == 0;                 // paired 16-row blocks: a lane holds 8 consecutive channels
  constexpr int WROWS = 16 * NB, WPIECES = WROWS / 8;   // filter rows / 1-KiB LDS-DMA pieces per stage
  constexpr int NWI = (WPIECES + 7) / 8;                // piece slots per wave (wave uw moves pieces uw + 8 i < WPIECES)
  constexpr int PPIECES = PROWS / 8, NPI = (PPIECES + 7) / 8;
  constexpr int WSLOT = WROWS * 128;
  constexpr int L = R - 1;                              // filter slices in flight
  constexpr int OFF_W = 2 * PSLOT, OFF_VEC = OFF_W + R * WSLOT;
  static_assert((L - 1) * NWI + NPI <= 16, "wait_vm_dyn range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wp = uw / WCH, wc = uw % WCH;
  const int wg = xcd_remap(blockIdx.x, p.ntiles);
  // (constant indices only: a run-time index into the by-value argument struct would move it to scratch memory)
  int sH = 1, sW = 1, stx = 1, stpi = 1, st0 = 0;
  long long ssrc = 0, sdst = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q < p.nseg && wg >= p.seg[q].tile0) {
      sH = p.seg[q].H; sW = p.seg[q].W; stx = p.seg[q].tiles_x; stpi = p.seg[q].tiles_per_img; st0 = p.seg[q].tile0;
      ssrc = p.seg[q].src0; sdst = p.seg[q].dst0;
    }
  const int local = wg - st0;
  const int b = local / stpi, rem = local - b * stpi;
  const int ty0 = (rem / stx) * TH, tx0 = (rem % stx) * TW;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  // the ReLU mask through a descriptor that is EMPTY when there is no mask (zeros come back, no branch around the loads)
  const auto rsrc_m = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask, 0, p.mask ? (int)(p.y_rows * p.N * 2) : 0, 0x00020000);

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
  int np_w = 0, nw_w = 0;                              // pieces THIS wave moves per halo chunk / filter slice (wave-uniform)
#pragma unroll
  for (int i = 0; i < NPI; ++i) np_w += (uw + 8 * i < PPIECES) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < NWI; ++i) nw_w += (uw + 8 * i < WPIECES) ? 1 : 0;
  unsigned poff[NPI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int row = 8 * (uw + 8 * i) + drow;
    const int hy = row / PW_, hx = row - hy * PW_;
    const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
    const bool ok = row < PPIX && (unsigned)y < (unsigned)sH && (unsigned)x < (unsigned)sW;
    poff[i] = ok ? (unsigned)((ssrc + ((long long)b * sH + y) * sW + x) * (long long)p.C * 2) : OOB;
  }
  auto issue_patch = [&](int kc, int slot) {
    const int ch = kc * 8 + kcl;
    const bool cok = ch * 8 < p.C;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (uw + 8 * i < PPIECES) {
        const unsigned off = (cok && poff[i] != OOB) ? poff[i] + (unsigned)(ch * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(smem + slot * PSLOT + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  // output pixels of this lane (one per pixel row block it owns)
  long long orow[NPW];
  bool pok[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int y = ty0 + wp * NPW + i, x = tx0 + lr;
    pok[i] = y < sH && x < sW;
    orow[i] = sdst + ((long long)b * sH + y) * sW + x;
  }
  constexpr int CW = PAIR ? 8 : 4;                      // consecutive channels a lane holds per unit
  constexpr int NU = PAIR ? NBW / 2 : NBW;
  const bool mvec = (p.N % CW) == 0;                    // mask / bf16 rows keep the vector alignment
  const bool nvec = mvec || (OUT_F32 && (p.N & 3) == 0);
  float* const csw = reinterpret_cast<float*>(smem);   // column-sum staging [WPX][256] floats over the (then dead) halo slots

  for (int nb0 = 0; nb0 * 16 < p.N; nb0 += NB) {
    if (nb0 > 0) __syncthreads();                       // every wave is out of the previous pass's K loop: its slots can be refilled
    unsigned wbase[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int n = nb0 * 16 + 8 * (uw + 8 * i) + drow;
      wbase[i] = n < p.N ? (unsigned)((long long)n * 9 * p.C * 2) : OOB;
    }
    auto issue_w = [&](int s, int slot) {
      const int kc = s / 9, tap = s - kc * 9;
      const int ch = kc * 8 + kcw;
      const bool cok = ch * 8 < p.C;
      const unsigned koff = (unsigned)((tap * p.C + ch * 8) * 2);
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        if (uw + 8 * i < WPIECES) {
          const unsigned off = (cok && wbase[i] != OOB) ? wbase[i] + koff : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + OFF_W + slot * WSLOT + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
        }
      }
    };
    // the producer's ReLU mask rows of this pass: requested ahead of the K loop, consumed after it
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x4_t mreg[NPW][NU];
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int jg0 = nb0 + wc * NBW + (PAIR ? 2 * u : u);
        const int ch0 = PAIR ? ((jg0 >> 1) * 32 + lq * 8) : jg0 * 16 + lq * 4;
        const unsigned off = (pok[i] && mvec && ch0 + CW <= p.N) ? (unsigned)((orow[i] * p.N + ch0) * 2) : OOB;
        if (PAIR) mreg[i][u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m, (int)off, 0, 0);
        else { const u32x2_t m2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc_m, (int)off, 0, 0); mreg[i][u] = (u32x4_t){m2[0], m2[1], 0u, 0u}; }
      }

    f32x4 acc[NPW][NBW];
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int j = 0; j < NBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_patch(0, 0);
#pragma unroll
    for (int s = 0; s < L; ++s)
      if (s < nsteps) issue_w(s, s);
    int kc = 0, tap = 0, slot = 0;
    for (int s = 0; s < nsteps; ++s) {
      // younger than this step's filter slice: the slices of the next L - 1 steps, and a halo chunk requested within the last L steps
      const int ahead = (nsteps - 1 - s) < (L - 1) ? (nsteps - 1 - s) : (L - 1);
      const bool pfly = tap >= 1 && tap <= L && kc + 1 < nsub;
      wait_vm_dyn(ahead * nw_w + (pfly ? np_w : 0));
      __builtin_amdgcn_s_barrier();               // every wave's pieces of this step have landed; every wave is done with step s - 1
      __builtin_amdgcn_sched_barrier(0);
      if (s + L < nsteps) issue_w(s + L, slot == 0 ? R - 1 : slot - 1);      // into the slot step s - 1 was read from
      if (tap == 0 && kc + 1 < nsub) issue_patch(kc + 1, (kc + 1) & 1);
      const char* ps = smem + (kc & 1) * PSLOT;
      const char* ws = smem + OFF_W + slot * WSLOT;
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
      slot = slot == R - 1 ? 0 : slot + 1;
    }

    // ---- epilogue straight from the accumulators: a lane holds CW consecutive channels of one pixel per unit
    float csum[NU][CW];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < CW; ++r) csum[u][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const bool ok = pok[i];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int jg0 = nb0 + wc * NBW + (PAIR ? 2 * u : u);
        const int ch0 = PAIR ? ((jg0 >> 1) * 32 + lq * 8) : jg0 * 16 + lq * 4;
        float v[CW];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][PAIR ? 2 * u : u][r] + vec[(ch0 + r) & 255];
          if (PAIR) v[4 + r] = acc[i][PAIR ? 2 * u + 1 : u][r] + vec[(ch0 + 4 + r) & 255];
        }
        const bool full = ch0 + CW <= p.N;
        if (p.mask && ok) {
          if (full && mvec) {
#pragma unroll
            for (int r = 0; r < CW; ++r) {
              const unsigned wd = mreg[i][u][r >> 1];
              const float m = __uint_as_float((r & 1) ? (wd & 0xffff0000u) : (wd << 16));
              v[r] = (m > 0.f) ? v[r] : 0.f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < CW; ++r)
              if (ch0 + r < p.N) v[r] = ((float)p.mask[orow[i] * p.N + ch0 + r] > 0.f) ? v[r] : 0.f;
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
            float* o = reinterpret_cast<float*>(p.y) + orow[i] * p.N + ch0;
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
            bf16_t* o = reinterpret_cast<bf16_t*>(p.y) + orow[i] * p.N + ch0;
            if (mvec && full) {
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
      // per channel and workgroup.  The staging area overlays the halo slots: every wave must be out of the K loop first.
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int jg0 = nb0 + wc * NBW + (PAIR ? 2 * u : u);
        const int ch0 = PAIR ? ((jg0 >> 1) * 32 + lq * 8) : jg0 * 16 + lq * 4;
#pragma unroll
        for (int r = 0; r < CW; ++r) {
          float v = csum[u][r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if (lr == 0) csw[wp * 256 + ((ch0 + r) & 255)] = v;
        }
      }
      __syncthreads();
      const int c = nb0 * 16 + t;
      if (t < 16 * NB && c < p.N) {
        float s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WPX; ++w) s2 += csw[w * 256 + (c & 255)];
        atomicAdd(p.colsum + c, s2);
      }
    }
  }
}

template <int NB, int WPX, int R, bool OUT_F32>
int launch_halo(const HaloArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * PSLOT + R * 16 * NB * 128 + 1024;
  static_assert(LDS <= 160 * 1024 && 2 * PSLOT >= WPX * 1024, "LDS map (the column-sum staging overlays the halo slots)");
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo_conv3x3_kernel<NB, WPX, R, OUT_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  hipLaunchKernelGGL((halo_conv3x3_kernel<NB, WPX, R, OUT_F32>), dim3(a.ntiles), dim3(512), LDS, st, a);
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
  // instances: one 16-channel block (retina_L), up to four (retina_reg) -- two workgroups per CU, three filter slices in flight --, up to
  // twelve in one pass (retina_cls), else passes of eight blocks with five slices in flight (the dgrads: N = 256)
  if (d->out_f32) {
    if (nb <= 1) launch_halo<1, 8, 4, true>(a, st);
    else if (nb <= 4) launch_halo<4, 4, 4, true>(a, st);
    else if (nb <= 12) launch_halo<12, 4, 4, true>(a, st);
    else launch_halo<8, 4, 6, true>(a, st);
  } else {
    if (nb <= 1) launch_halo<1, 8, 4, false>(a, st);
    else if (nb <= 4) launch_halo<4, 4, 4, false>(a, st);
    else if (nb <= 12) launch_halo<12, 4, 4, false>(a, st);
    else launch_halo<8, 4, 6, false>(a, st);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}
