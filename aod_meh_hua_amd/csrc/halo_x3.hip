// The NARROW prediction convs of the head in the reference-precision mode (conv.hip "X3"): 3x3 / stride-1 / pad-1 over the five pyramid levels,
// 256 -> 36 (retina_reg) and 256 -> 9 (retina_L) output channels, fp32 NHWC output (mmdet/models/dense_heads/Lambda_L2.py:52-54,92-103).
// The general implicit-GEMM kernel computes 64 output columns for them (44 % / 86 % of its matrix work is padding) and gathers every tap's rows
// again: 111 / 100 us per launch at 16 x 512^2.  Here one 8-wave workgroup owns an 8 x 16 pixel tile of one (level, image), like halo_conv.hip,
// but a K-STEP IS A WHOLE 32-CHANNEL CHUNK: the 10 x 18 halo of the chunk (184 rows x 128 B of X rows: heads | tails) AND the chunk's filter
// slices of all nine taps ([9][16 NB rows][128 B]) arrive together by LDS-DMA, so the nine taps run back to back between two barriers -- eight
// barriers per tile where the bf16 halo kernel has seventy-two --, and the next chunk(s) stream in meanwhile.  K order = (chunk, tap) with the tap
// innermost and per step heads x heads, activation tails x filter heads, activation heads x filter tails: the order conv_igemm_kernel<X3> walks
// these layers in (C >= 256 columns: taps innermost) -- identical bits.  Wave w = output pixel row w of the tile; products are computed transposed
// (filter fragment = the MFMA's A operand), so a lane ends up with 4 consecutive output channels of one pixel: 16-B fp32 stores straight from
// the accumulators, bias and ReLU fused.
#include "common.h"

namespace {

struct Hx3Seg { int B, H, W, tiles_x, tiles_per_img, tile0; long long src0, dst0; };
struct Hx3Args {
  const bf16_t* x;       // X rows [rows][C]  (C = physical width: 2 * ceil32(logical channels))
  const bf16_t* w;       // X filter [N][3][3][C]
  float* y;              // [rows][N] fp32
  const float* shift;    // [N] or null
  int C, N, relu, nseg, ntiles;
  long long x_bytes, w_bytes;
  Hx3Seg seg[8];
};

constexpr int TH = 8, TW = 16, PW_ = TW + 2, PPIX = (TH + 2) * PW_;      // 180 halo pixels
constexpr int PROWS = 184, PPIECES = PROWS / 8;                            // 23 LDS-DMA pieces of 8 rows (waves 0-6 move three, wave 7 two)
constexpr int PSLOT = PROWS * 128;                                        // one 32-channel chunk of the halo: 23 552 B
constexpr unsigned OOB = 0xf0000000u;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define AOD_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    AOD_VMCASE(0) AOD_VMCASE(1) AOD_VMCASE(2) AOD_VMCASE(3) AOD_VMCASE(4) AOD_VMCASE(5) AOD_VMCASE(6) AOD_VMCASE(7) AOD_VMCASE(8)
    AOD_VMCASE(9) AOD_VMCASE(10) AOD_VMCASE(11) AOD_VMCASE(12) AOD_VMCASE(13) AOD_VMCASE(14) AOD_VMCASE(15) AOD_VMCASE(16)
    AOD_VMCASE(17) AOD_VMCASE(18) AOD_VMCASE(19) AOD_VMCASE(20) AOD_VMCASE(21) AOD_VMCASE(22) AOD_VMCASE(23) AOD_VMCASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef AOD_VMCASE

// NB: 16-channel output blocks (N <= 16 NB); S: stages (chunks resident / in flight); RW: output pixel rows per wave (8 / RW waves).
// RW = 1 (default): every wave reads all of a chunk's filter fragments for its one row -- 72 ds_read_b128 per 81 MFMAs at NB = 3.  RW = 2 (four
// waves): a wave reads the filter fragments once for two rows and the halo fragments of its four halo rows once per chunk (a tap (dy, dx) of
// row r is halo row r + dy: 12 fragments instead of 18), 78 reads per 162 MFMAs.  Same (chunk, tap, product) order per accumulator: identical
// bits -- and the same time: what bounds the kernel is the arrival of its chunks (77 KB each at NB = 3, one ahead of the one in use), not the LDS pipe.
template <int NB, int S, int RW>
__global__ __launch_bounds__(512 / RW) __attribute__((amdgpu_waves_per_eu(1, 2))) void halo_x3_kernel(const Hx3Args p) {
  constexpr int NWV = 8 / RW;                                             // waves
  constexpr int WROWS = 16 * NB, WPIECES = 9 * WROWS / 8;                 // filter rows per tap; LDS-DMA pieces per chunk (all nine taps)
  constexpr int NPI = (PPIECES + NWV - 1) / NWV, NWI = (WPIECES + NWV - 1) / NWV;         // piece slots per wave
  constexpr int STAGE = PSLOT + 9 * WROWS * 128;
  constexpr int OFF_VEC = S * STAGE;
  static_assert(RW == 1 || RW == 2, "rows per wave");
  static_assert(OFF_VEC + 64 * 4 <= 160 * 1024, "LDS map");
  static_assert((S - 1) * (NPI + NWI) <= 24, "wait_vm_dyn range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
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

  // bias vector -> LDS (an ordinary global load beside LDS-DMA makes hipcc drain the DMA queue where the value is used: do it first)
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if (t < 64) vec[t] = (p.shift && t < p.N) ? p.shift[t] : 0.f;

  // LDS-DMA lane roles: one wave-instruction fills 8 rows x 8 slots; lane -> row (lane >> 3) of its piece, slot lane & 7, source chunk
  // slot ^ key(row); a wave moves the pieces uw + NWV i of an image, for which the key is the same (NWV is even)
  const int drow = lane >> 3;
  const int kcl = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int nsub = p.C >> 6;                                              // 32-channel chunks
  int n_w = 0;                                                            // pieces THIS wave moves per chunk (wave-uniform)
#pragma unroll
  for (int i = 0; i < NPI; ++i) n_w += (uw + NWV * i < PPIECES) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < NWI; ++i) n_w += (uw + NWV * i < WPIECES) ? 1 : 0;
  unsigned poff[NPI], woff[NWI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int row = 8 * (uw + NWV * i) + drow;
    const int hy = row / PW_, hx = row - hy * PW_;
    const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
    const bool ok = row < PPIX && (unsigned)y < (unsigned)sH && (unsigned)x < (unsigned)sW;
    poff[i] = ok ? (unsigned)((ssrc + ((long long)b * sH + y) * sW + x) * (long long)p.C * 2 + kcl * 16) : OOB;
  }
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int pi = uw + NWV * i;                                          // piece of the chunk's filter image [9][WROWS][128 B]
    const int tap = pi / (WROWS / 8), n = (pi - tap * (WROWS / 8)) * 8 + drow;
    woff[i] = (pi < WPIECES && n < p.N) ? (unsigned)((((long long)n * 9 + tap) * p.C) * 2 + kcl * 16) : OOB;
  }
  auto issue = [&](int kc, int slot) {                                    // chunk kc -> stage `slot`: its halo and its nine filter slices
    char* st = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < NPI; ++i)
      if (uw + NWV * i < PPIECES) {
        const unsigned off = poff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(st + (uw + NWV * i) * 1024), 16, off, kc * 128, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < NWI; ++i)
      if (uw + NWV * i < WPIECES) {
        const unsigned off = woff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(st + PSLOT + (uw + NWV * i) * 1024), 16, off, kc * 128, 0, 0);
      }
  };
  // per-lane fragment addresses inside a stage (heads; the tails sit at slot ^ 4 = 64 B further): halo row RW uw + hr, pixel lr + dx
  unsigned afrag[RW + 2][3], wfrag[NB];
#pragma unroll
  for (int hr = 0; hr < RW + 2; ++hr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) afrag[hr][dx] = (unsigned)swz((RW * uw + hr) * PW_ + lr + dx, lq);
#pragma unroll
  for (int j = 0; j < NB; ++j) wfrag[j] = (unsigned)(PSLOT + swz(j * 16 + lr, lq));
  auto lds16 = [&](unsigned a) { return *reinterpret_cast<const bf16x8*>(smem + a); };

  f32x4 acc[RW][NB];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[rr][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (s < nsub) issue(s, s);
  int slot = 0;
  for (int kc = 0; kc < nsub; ++kc) {
    // chunks issued so far: 0 .. S - 1 ahead of the loop, chunk c - 1 + S at iteration c >= 1; the younger ones may stay in flight
    const int last = kc == 0 ? S - 1 : kc - 2 + S;
    const int young = (last < nsub - 1 ? last : nsub - 1) - kc;
    wait_vm_dyn((young > 0 ? young : 0) * n_w);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of chunk kc - 1 are complete (kc == 0: its vec[] store)
    __builtin_amdgcn_s_barrier();                 // chunk kc has landed for every wave; every wave is done with chunk kc - 1
    __builtin_amdgcn_sched_barrier(0);
    if (kc >= 1 && kc - 1 + S < nsub) issue(kc - 1 + S, slot == 0 ? S - 1 : slot - 1);      // into the stage chunk kc - 1 was read from
    __builtin_amdgcn_sched_barrier(0);
    const unsigned base = (unsigned)(slot * STAGE);
    if constexpr (RW == 1) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const unsigned af = afrag[tap / 3][tap % 3];
        const bf16x8 ah = lds16(base + af), al = lds16(base + (af ^ 64u));
        bf16x8 wh[NB], wl[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          wh[j] = lds16(base + tap * (WROWS * 128) + wfrag[j]);
          wl[j] = lds16(base + tap * (WROWS * 128) + (wfrag[j] ^ 64u));
        }
        // (per accumulator the order of conv_igemm_kernel<X3>: heads x heads, activation tails x filter heads, activation heads x filter tails)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc[0][j], 0, 0, 0);
      }
    } else {
      // the chunk's halo fragments of this wave's RW + 2 halo rows, once
      bf16x8 ah[RW + 2][3], al[RW + 2][3];
#pragma unroll
      for (int hr = 0; hr < RW + 2; ++hr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) { ah[hr][dx] = lds16(base + afrag[hr][dx]); al[hr][dx] = lds16(base + (afrag[hr][dx] ^ 64u)); }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        bf16x8 wh[NB], wl[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          wh[j] = lds16(base + tap * (WROWS * 128) + wfrag[j]);
          wl[j] = lds16(base + tap * (WROWS * 128) + (wfrag[j] ^ 64u));
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[rr][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[rr + dy][dx], acc[rr][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[rr][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[rr + dy][dx], acc[rr][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[rr][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[rr + dy][dx], acc[rr][j], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    slot = slot == S - 1 ? 0 : slot + 1;
  }

  // ---- epilogue straight from the accumulators: lane (lr, lq) holds channels 16 j + 4 lq .. + 3 of pixel (ty0 + RW uw + rr, tx0 + lr)
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int y = ty0 + RW * uw + rr, x = tx0 + lr;
    if (y < sH && x < sW) {
      float* o = p.y + (sdst + ((long long)b * sH + y) * sW + x) * p.N;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int ch0 = j * 16 + lq * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[rr][j][r] * 1.0f + vec[(ch0 + r) & 63];          // (the general kernel's epilogue: v * scale + shift with scale = 1)
          if (p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        if (ch0 + 4 <= p.N && (p.N & 3) == 0) *reinterpret_cast<f32x4*>(o + ch0) = (f32x4){v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ch0 + r < p.N) o[ch0 + r] = v[r];
        }
      }
    }
  }
}

template <int NB, int S, int RW>
int launch_hx3(const Hx3Args& a, hipStream_t st) {
  constexpr int LDS = S * (PSLOT + 9 * 16 * NB * 128) + 256;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo_x3_kernel<NB, S, RW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipLaunchKernelGGL((halo_x3_kernel<NB, S, RW>), dim3(a.ntiles), dim3(512 / RW), LDS, st, a);
  return 0;
}

}  // namespace

// 1 when aod_halo_conv3x3_x3 handles this descriptor: an x3 forward 3x3 / stride 1 / pad 1 conv with an fp32 destination of at most 48 channels
extern "C" int aod_halo_conv3x3_x3_applies(const aod_conv_desc_t* d) {
  if (!d || !d->x3 || d->transposed || !d->out_f32 || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return 0;
  if (d->nseg < 1 || d->nseg > 8 || d->C % 64 != 0 || d->C < 64 || d->N < 1 || d->N > 48) return 0;
  for (int i = 0; i < d->nseg; ++i)
    if (d->seg[i].OH != d->seg[i].H || d->seg[i].OW != d->seg[i].W) return 0;
  return 1;
}

extern "C" int aod_halo_conv3x3_x3(const aod_conv_desc_t* d, const void* src, const void* w_packed, void* dst, const float* pre_shift,
                                   aod_stream_t stream) {
  AOD_CHECK_ARG(d && src && w_packed && dst, "halo_conv_x3: null pointer");
  AOD_CHECK_ARG(aod_halo_conv3x3_x3_applies(d), "halo_conv_x3: needs an x3 forward 3x3 / stride-1 / pad-1 conv, fp32 destination, N <= 48");
  Hx3Args a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)src; a.w = (const bf16_t*)w_packed; a.y = (float*)dst; a.shift = pre_shift;
  a.C = d->C; a.N = d->N; a.relu = d->relu; a.nseg = d->nseg;
  long long xrows = 0;
  int tiles = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const aod_conv_seg_t& s = d->seg[i];
    AOD_CHECK_ARG(s.src_row0 >= 0 && s.dst_row0 >= 0 && s.B >= 0, "halo_conv_x3: negative row offset");
    Hx3Seg& h = a.seg[i];
    h.B = s.B; h.H = s.H; h.W = s.W; h.src0 = s.src_row0; h.dst0 = s.dst_row0;
    h.tiles_x = (s.W + TW - 1) / TW;
    h.tiles_per_img = h.tiles_x * ((s.H + TH - 1) / TH);
    h.tile0 = tiles;
    tiles += s.B * h.tiles_per_img;
    const long long e = s.src_row0 + (long long)s.B * s.H * s.W;
    if (e > xrows) xrows = e;
  }
  for (int i = d->nseg; i < 8; ++i) a.seg[i].tile0 = 0x7fffffff;
  if (tiles == 0) return 0;
  a.ntiles = tiles;
  a.x_bytes = xrows * a.C * 2;
  a.w_bytes = (long long)a.N * 9 * a.C * 2;
  AOD_CHECK_ARG(a.x_bytes < 0xe0000000ll && a.w_bytes < 0xe0000000ll, "halo_conv_x3: operand larger than 3.5 GiB (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  // (read per call: tests switch it in-process) AOD_HALO_X3_RW=2: two pixel rows per wave, four waves -- half the LDS fragment reads, the same
  // time (84 / 48 us either way, profiles/r05_x3_tile_ab.txt): the kernel waits for its chunks, one or two of which fit the LDS ahead of the one in use
  const char* e = getenv("AOD_HALO_X3_RW");
  const bool rw2 = e && e[0] == '2';
  if (a.N <= 16) { if (rw2) launch_hx3<1, 3, 2>(a, st); else launch_hx3<1, 3, 1>(a, st); }      // retina_L: 41 KB per chunk, three stages
  else { if (rw2) launch_hx3<3, 2, 2>(a, st); else launch_hx3<3, 2, 1>(a, st); }                // retina_reg: 77 KB per chunk, two stages
  AOD_LAUNCH_CHECK();
  return 0;
}
