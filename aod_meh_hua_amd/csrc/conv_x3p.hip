// Reference-precision (x3) convolution as a PERSISTENT PRODUCER / CONSUMER kernel for gfx950 -- the short-K / small-M layers of the backbone
// and the neck (mmdet/models/backbones/resnet.py:262-301, necks/fpn.py:151-202), forward and stride-1 dgrad, 1x1 and 3x3.
//
// Why a second kernel.  conv_igemm_kernel (conv.hip) is one program per wave: every wave resolves the im2col addresses of its share of the
// next K-step (run-time tap / stride / class / segment cases: ~130 - 500 scalar and vector instructions per K-step in the generated code),
// issues its LDS-DMA, reads fragments and issues 48 MFMAs, with a __syncthreads (vmcnt(0)) per K-step.  On the layers with 256 - 512 tiles
// (one or two 4-wave workgroups per CU) nothing covers the address code and the load issue: the matrix pipe is busy 27 % of the time
// (profiles/r05_pmc_passes.txt), although neither the L2 -> LDS path nor the LDS is near a limit.
//
// Structure here (one 8-wave workgroup per CU, 128 pixels x 128 channels per tile, grid = min(tiles, CUs), each workgroup walks its tiles):
//   * waves 4-7 are LOADERS: per tile they resolve 4 pixel rows per lane ONCE into one byte offset per (tap, row) -- out-of-image taps become
//     an out-of-range offset, for which the buffer range check returns zeros -- and then only issue LDS-DMA: 8 x 1 KiB per wave and K-step,
//     the K-step's displacement in the instruction's scalar offset.  They keep three 32-KB stages in flight in a five-slot ring (all 160 KB of
//     the LDS) and continue into the NEXT tile while the consumers finish the current one (the ring does not know about tiles).
//   * waves 0-3 are CONSUMERS (one per SIMD, 2 x 2 over the tile): fragment reads + 48 MFMAs per K-step, nothing else.  Products are formed
//     TRANSPOSED -- the filter is the MFMA's A operand, the pixels its B operand -- and the loaders fill the filter tile with its rows
//     permuted, so that a lane ends up with eight consecutive channels of one pixel in two accumulators: the epilogue (BN / bias, residual =
//     head + tail, ReLU mask, ReLU, column sums, head / tail split) runs in registers and stores 16-B pieces straight to the destination
//     rows -- no LDS image, no barrier, and the ring keeps filling meanwhile.
//   * one raw s_barrier per K-step.  At barrier g the loaders guarantee that stage g + 1 has landed (counted vmcnt: the 24 instructions of
//     the three younger stages may stay in flight) and the consumers that their reads of stage g are complete (stage g + 5 takes its slot).
//
// Same products in the same order per accumulator as conv_igemm_kernel<.., X3 = true> (K-steps in (chunk, tap) order for C >= 256, (tap, chunk)
// otherwise; xh*wh, xl*wh, xh*wl per step) and the same fp32 epilogue arithmetic: IDENTICAL BITS (tests/test_gpu_x3p.py); only the fp32
// atomics of the optional column sums arrive in another order.
#include "conv_x3p.h"

namespace {

constexpr int XP_BM = 128, XP_BN = 128, XP_STAGE = 32768, XP_ABYTES = 16384;
// ring: XP_RING slots, XP_FLY stages in flight.  Iteration k of a loader issues stage k, waits until all but the youngest XP_FLY stages'
// instructions have landed (stage k - XP_FLY and older) and joins barrier k - XP_FLY - 1, behind which the consumers read stage k - XP_FLY;
// stage k overwrites the slot of stage k - XP_RING, whose reads ended before barrier k - XP_RING <= k - XP_FLY - 2: XP_RING >= XP_FLY + 2.
constexpr int XP_RING = 5, XP_FLY = 3;
static_assert(XP_RING >= XP_FLY + 2, "ring depth");
constexpr unsigned XP_OOB = 0xf0000000u;

__device__ __forceinline__ unsigned xp_udiv(unsigned n, unsigned d) {      // n / d for n < 2^22 (conv.hip udiv_small)
  unsigned q = (unsigned)((float)n * __builtin_amdgcn_rcpf((float)d));
  const int r = (int)(n - q * d);
  q = r < 0 ? q - 1 : ((unsigned)r >= d ? q + 1 : q);
  return q;
}

__device__ __forceinline__ int xp_swizzle(int bid, int nwg) {      // bijective: blocks that share an XCD (bid % 8) take a contiguous tile range
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <class T>
__device__ __forceinline__ T xp_sel8(const T* a, int sg) {      // a[sg] of a by-value argument array with constant indices only (sg is wave-uniform)
  T v = a[0];
#pragma unroll
  for (int q = 1; q < 8; ++q) v = sg == q ? a[q] : v;
  return v;
}

template <int TAPS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_x3p_kernel(const X3PArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;
  const int nmine = (ntiles - bid + G - 1) / G;              // tiles bid, bid + G, ... (launcher: G <= ntiles)
  const int CC = p.C >> 6;                                    // 64-column (= 32-channel) chunks per tap
  const int nk = TAPS * CC;                                   // K-steps per tile
  const int total = nmine * nk;
  auto tile_of = [&](int j) {
    const int base = j * G, left = ntiles - base;
    return base + xp_swizzle(bid, left < G ? left : G);
  };
  auto seg_of = [&](int m) {
    int sg = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) sg += (q + 1 < p.nseg && m >= p.seg_mend[q]) ? 1 : 0;
    return sg;
  };

  if (wave >= 4) {
    // ======================================================================== loaders
    const int w = wave - 4;
    const int prow = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);          // k-chunk this lane fetches: slot ^ ((row >> 1) & 7), the same for its 4 rows
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    // filter rows: LDS row R of the tile holds output channel n0 + perm(R), perm = the order in which the transposed product leaves eight
    // consecutive channels in one lane (block pair a = 2 x 16 rows -> channels 32a + 8q + {0..3 | 4..7}, see the consumers' epilogue)
    unsigned wvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int R = 8 * w + 32 * i + prow;
      const int nrel = (R >> 6) * 64 + ((R >> 5) & 1) * 32 + ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3);
      wvo[i] = (unsigned)((nrel * p.K + kc * 8) * 2);
    }
    const int sgn = p.transposed ? -1 : 1;
    int gi = 0, slot = 0;                                      // stages issued so far, ring slot of the next one
    char* const lds_w = smem + (8 * w) * 128;
    for (int j = 0; j < nmine; ++j) {
      const int tile = tile_of(j);
      const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
      const int m0 = tm * XP_BM, n0 = tn * XP_BN;
      // the tile lies inside ONE segment (launcher): workgroup-uniform geometry
      const int sg = seg_of(m0);
      const unsigned mstart = sg ? (unsigned)xp_sel8(p.seg_mend, sg - 1) : 0u;
      const unsigned OW = (unsigned)xp_sel8(p.segOW, sg), OH = (unsigned)xp_sel8(p.segOH, sg), ohw = OH * OW;
      const unsigned H = (unsigned)xp_sel8(p.segH, sg), W = (unsigned)xp_sel8(p.segW, sg);
      const unsigned rowb = (unsigned)(p.C * 2);
      const unsigned src0b = (unsigned)((unsigned long long)xp_sel8(p.seg_src0, sg) * rowb);       // (32-bit byte offsets: x_bytes < 3.5 GiB)
      const unsigned imgb = H * W * rowb;
      unsigned vo[TAPS][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * w + 32 * i + prow;
        const bool live = m < p.M;
        const unsigned ml = (unsigned)m - mstart;
        const unsigned b = xp_udiv(ml, ohw), rem = ml - b * ohw, oy = xp_udiv(rem, OW), ox = rem - oy * OW;
        const int y0 = p.transposed ? (int)oy + p.pad : (int)oy * p.stride - p.pad;
        const int x0 = p.transposed ? (int)ox + p.pad : (int)ox * p.stride - p.pad;
        const unsigned base = src0b + b * imgb + (unsigned)(kc * 16);
        constexpr int RR = TAPS == 9 ? 3 : 1;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
          const int y = y0 + sgn * r * p.dil;
          const bool yok = live && (unsigned)y < H;
#pragma unroll
          for (int s = 0; s < RR; ++s) {
            const int x = x0 + sgn * s * p.dil;
            const bool ok = yok && (unsigned)x < W;
            vo[r * RR + s][i] = ok ? base + (unsigned)(y * (int)W + x) * rowb : XP_OOB;
          }
        }
      }
      const unsigned soff_w0 = (unsigned)n0 * (unsigned)(p.K * 2);
      auto issue = [&](const unsigned (&v)[4], int t, int cc) {
        char* const sa = lds_w + slot * XP_STAGE;
        const unsigned soff_x = (unsigned)cc * 128u;
        const unsigned soff_w = soff_w0 + (unsigned)((t * p.C + cc * 64) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned off = v[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sa + i * 4096), 16, off, soff_x, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned off = wvo[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sa + XP_ABYTES + i * 4096), 16, off, soff_w, 0, 0);
        }
        ++gi;
        slot = slot == XP_RING - 1 ? 0 : slot + 1;
        if (gi > XP_FLY) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * XP_FLY) : "memory");
          __builtin_amdgcn_s_barrier();
        }
      };
      if (TAPS == 1 || p.tapin) {
        // (p.rot, timing experiments only: the chunk loop starts at a tile-dependent chunk -- another summation order)
        int cc = p.rot ? (tm * p.rot) % CC : 0;
        for (int c0 = 0; c0 < CC; ++c0) {
#pragma unroll
          for (int t = 0; t < TAPS; ++t) issue(vo[t], t, cc);
          cc = cc + 1 == CC ? 0 : cc + 1;
        }
      } else {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
          for (int cc = 0; cc < CC; ++cc) issue(vo[t], t, cc);
      }
    }
    // the barriers of the last stages: total + 1 in all (one opens the ring, one closes every K-step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int done = total > XP_FLY ? total - XP_FLY : 0;
    for (int k = done; k < total + 1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }

  // ========================================================================== consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int sw = (lr >> 1) & 7;
  const int oh = (lq ^ sw) << 4, ol = ((4 + lq) ^ sw) << 4;          // 16-B slot of this lane's head / tail k-chunk inside a 128-B tile row
  const int xoff = (wm * 64 + lr) * 128, woff = XP_ABYTES + (wn * 64 + lr) * 128;
  const int NP = ((p.N + 31) >> 5) << 6;                             // destination row pitch (X-layout, elements)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_shift, 0, p.pre_shift ? p.N * 4 : 0, 0x00020000);     // empty descriptor -> zeros
  const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_scale, 0, p.pre_scale ? p.N * 4 : 0, 0x00020000);

  float csum[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 8; ++q) csum[a][q] = 0.f;
  auto flush_colsum = [&](int n0) {
    // the 16 lanes of a quarter (same lq) hold the same 16 channels: butterfly over lr, then lane lr sends channel (lr >> 3, lr & 7) of its
    // quarter -- one atomic wave-instruction with 64 distinct, contiguous words
    float mine = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float s = csum[a][q];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        mine = lr == a * 8 + q ? s : mine;
        csum[a][q] = 0.f;
      }
    atomicAdd(p.colsum + n0 + wn * 64 + 32 * (lr >> 3) + 8 * lq + (lr & 7), mine);
  };
  __builtin_amdgcn_s_barrier();                                      // stage 0 is resident
  int g = 0, cslot = 0;
  for (int j = 0; j < nmine; ++j) {
    const int tile = tile_of(j);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * XP_BM, n0 = tn * XP_BN;
    f32x4 acc[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nk; ++ks, ++g) {
      const char* const st = smem + cslot * XP_STAGE;
      cslot = cslot == XP_RING - 1 ? 0 : cslot + 1;
      bf16x8 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wh[q] = *reinterpret_cast<const bf16x8*>(st + woff + oh + q * 2048);
        xh[q] = *reinterpret_cast<const bf16x8*>(st + xoff + oh + q * 2048);
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nb], xh[mb], acc[nb][mb], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) xl[q] = *reinterpret_cast<const bf16x8*>(st + xoff + ol + q * 2048);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nb], xl[mb], acc[nb][mb], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) wl[q] = *reinterpret_cast<const bf16x8*>(st + woff + ol + q * 2048);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nb], xh[mb], acc[nb][mb], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue, in registers.  acc[2a][mb][r] = channel cb + r, acc[2a + 1][mb][r] = channel cb + 4 + r of pixel m0 + wm * 64 + mb * 16 + lr,
    // cb = n0 + wn * 64 + 32a + 8 * lq: head columns 2 * (cb - 8 lq) + 8 lq .. + 7 of the destination row, tails 32 columns further
    const int sg = seg_of(m0);
    const long long drow0 = xp_sel8(p.seg_dst0, sg) - (long long)(sg ? xp_sel8(p.seg_mend, sg - 1) : 0);
    float cs1[2][8], cb1[2][8];
    bf16x8 rh[2][4], rl[2][4], mh[2][4];
    long long eoff[2][4];
    bool live[4];
    // every operand of the tile is requested before the first one is used (two halves x four pixel blocks: 16 + 8 16-B loads per lane)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = n0 + wn * 64 + 32 * a + 8 * lq;
      const u32x4_t b0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, cb * 4, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, cb * 4 + 16, 0, 0);
      const u32x4_t s0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, cb * 4, 0, 0), s1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, cb * 4 + 16, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cb1[a][q] = __uint_as_float(b0[q]); cb1[a][4 + q] = __uint_as_float(b1[q]);
        cs1[a][q] = __uint_as_float(s0[q]); cs1[a][4 + q] = __uint_as_float(s1[q]);
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int m = m0 + wm * 64 + mb * 16 + lr;
        live[mb] = m < p.M;
        // (rows past M read the last row's operands instead of branching per lane around the loads: a divergent branch per load makes the
        // compiler wait for each one where the paths merge; only the stores are predicated)
        eoff[a][mb] = (drow0 + (live[mb] ? m : p.M - 1)) * NP + 2 * (cb - 8 * lq) + 8 * lq;
        if (p.res) {
          rh[a][mb] = *reinterpret_cast<const bf16x8*>(p.res + eoff[a][mb]);
          rl[a][mb] = *reinterpret_cast<const bf16x8*>(p.res + eoff[a][mb] + 32);
        }
        if (p.mask) mh[a][mb] = *reinterpret_cast<const bf16x8*>(p.mask + eoff[a][mb]);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if (!live[mb]) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = acc[2 * a][mb][q]; v[4 + q] = acc[2 * a + 1][mb][q]; }
        if (p.pre_scale) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] *= cs1[a][q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += cb1[a][q];
        if (p.res) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += (float)rh[a][mb][q] + (float)rl[a][mb][q];
        }
        if (p.mask) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = ((float)mh[a][mb][q] > 0.f) ? v[q] : 0.f;
        }
        if (p.relu) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        bf16x8 ov, ovl;
#pragma unroll
        for (int q = 0; q < 8; ++q) { csum[a][q] += v[q]; ov[q] = (bf16_t)v[q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) ovl[q] = (bf16_t)(v[q] - (float)ov[q]);
        *reinterpret_cast<bf16x8*>(p.y + eoff[a][mb]) = ov;
        *reinterpret_cast<bf16x8*>(p.y + eoff[a][mb] + 32) = ovl;
      }
    }
    // column sums: the lane keeps adding its pixels' values for its 16 channels over ALL tiles of this workgroup that share the column tile
    // (nearly always all of them: the grid stride is a multiple of tiles_n), and sends them once -- 64 four-lane atomics per TILE into the
    // same 256 words from every CU ran the layer-3 dgrad 28 % slower than the general kernel's two full-wave atomics per tile
    if (p.colsum) {
      const int tn_next = j + 1 < nmine ? (tile_of(j + 1) % p.tiles_n) : -1;
      if (tn_next != tn) flush_colsum(n0);
    }
  }
}

int g_x3p_cus[64];
long long g_x3p_count = 0;

}  // namespace

int aod_conv_x3p_wants(const X3PArgs& a, int deterministic_colsum) {
  static const char* dbg = getenv("AOD_X3P");
  const char* now = getenv("AOD_X3P");                 // (read per call: tests switch it in-process)
  (void)dbg;
  if (now && now[0] == '0') return 0;
  if (a.taps != 1 && a.taps != 9) return 0;
  // 1x1 layers stay with the general kernel unless AOD_X3P_1X1=1: every K-step of theirs needs pixel bytes from beyond the L2 (nothing is
  // re-read tap after tap), both kernels then run at the ~33 GB/s per CU that path delivers and the persistent form gains nothing
  // (tools/dbg/x3p_micro.py, profiles/r06_x3p_micro.txt: 38 - 40 us either way on the stage-3 reduce conv, 50 vs 47 us on its expand conv)
  { const char* pw = getenv("AOD_X3P_1X1"); if (a.taps == 1 && !(pw && pw[0] == '1')) return 0; }
  if (a.taps == 9 && a.S != 3) return 0;
  if (a.N % XP_BN != 0 || a.C % 64 != 0 || a.M <= 0) return 0;
  if (a.transposed && a.stride != 1) return 0;
  if (a.colsum && deterministic_colsum) return 0;      // ordered column sums stay with the general kernel (determinism.hip)
  if ((long long)a.K * 2 * a.N >= 0x7fffffffll) return 0;
  long long prev = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const long long rows = a.seg_mend[i] - prev;
    prev = a.seg_mend[i];
    if (rows >= (1ll << 22)) return 0;                               // float-reciprocal row decode
    if (i + 1 < a.nseg && rows % XP_BM != 0) return 0;               // a tile must not straddle two segments
    if (a.seg_dst0[i] + rows >= (1ll << 31)) return 0;
  }
  // one 8-wave workgroup per CU: worth it from ~ a round of the chip on; below that the general kernel's smaller tiles fill more CUs
  const char* mint = getenv("AOD_X3P_MIN_TILES");
  const long long tiles = (long long)((a.M + XP_BM - 1) / XP_BM) * (a.N / XP_BN);
  if (tiles < (mint ? atoll(mint) : 192)) return 0;
  return 1;
}

int aod_conv_x3p_launch(const X3PArgs& a0, hipStream_t st) {
  X3PArgs a = a0;
  a.tiles_m = (a.M + XP_BM - 1) / XP_BM;
  a.tiles_n = a.N / XP_BN;
  { const char* r = getenv("AOD_X3P_ROT"); a.rot = r ? atoi(r) : 0; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  if (!g_x3p_cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_x3p_cus[dev] = n;
  }
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = ntiles < g_x3p_cus[dev] ? ntiles : g_x3p_cus[dev];
  const size_t lds = (size_t)XP_RING * XP_STAGE;
  static unsigned long long attr1 = 0, attr9 = 0;
  if (a.taps == 1) {
    if (aod_first_on_device(&attr1)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv_x3p_kernel<1>, dim3(grid), dim3(512), lds, st, a);
  } else {
    if (aod_first_on_device(&attr9)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv_x3p_kernel<9>, dim3(grid), dim3(512), lds, st, a);
  }
  AOD_LAUNCH_CHECK();
  __atomic_add_fetch(&g_x3p_count, 1, __ATOMIC_RELAXED);
  return 0;
}

extern "C" int64_t aod_conv_x3p_count(void) { return (int64_t)__atomic_load_n(&g_x3p_count, __ATOMIC_RELAXED); }
