// Reference-precision (x3) convolution as a PERSISTENT PRODUCER / CONSUMER kernel for gfx950 -- the short-K / small-M layers of the backbone
// and the neck (mmdet/models/backbones/resnet.py:262-301, necks/fpn.py:151-202), forward and stride-1 dgrad, 1x1 and 3x3.
//
// Why a second kernel.  conv_igemm_kernel (conv.hip) is one program per wave: every wave resolves the im2col addresses of its share of the
// next K-step (run-time tap / stride / class / segment cases: ~130 - 500 scalar and vector instructions per K-step in the generated code),
// issues its LDS-DMA, reads fragments and issues 48 MFMAs, with a __syncthreads (vmcnt(0)) per K-step.  On the layers with 256 - 512 tiles
// (one or two 4-wave workgroups per CU) nothing covers the address code and the load issue: the matrix pipe is busy 27 % of the time
// (profiles/r05_pmc_passes.txt), although neither the L2 -> LDS path nor the LDS is near a limit.
//
// Structure here (one 8-wave workgroup per CU, 128 pixels x 128 or 256 channels per tile, grid = min(tiles, CUs), each workgroup walks its tiles):
//   * waves 4-7 are LOADERS: per tile they resolve 4 pixel rows per lane ONCE into one byte offset per (tap, row) -- out-of-image taps become
//     an out-of-range offset, for which the buffer range check returns zeros -- and then only issue LDS-DMA: 8 x 1 KiB per wave and K-step,
//     the K-step's displacement in the instruction's scalar offset.  They keep three 32-KB stages in flight in a five-slot ring (128-column
//     tiles; 256-column tiles: one to two 48-KB stages in a three-slot ring -- either way all but 16 KB of the LDS) and continue into the NEXT
//     tile while the consumers finish the current one (the ring does not know about tiles).
//   * waves 0-3 are CONSUMERS (one per SIMD, 2 x 2 over the tile): fragment reads + 48 MFMAs per K-step, nothing else.  Products are formed
//     TRANSPOSED -- the filter is the MFMA's A operand, the pixels its B operand -- and the loaders fill the filter tile with its rows
//     permuted, so that a lane ends up with eight consecutive channels of one pixel in two accumulators: the epilogue (BN / bias, residual =
//     head + tail, ReLU mask, ReLU, column sums, head / tail split) runs in registers and stores 16-B pieces straight to the destination
//     rows -- no LDS image, no barrier, and the ring keeps filling meanwhile.
//   * one raw s_barrier per K-step.  At barrier g the loaders guarantee that stage g + 1 has landed (counted vmcnt: the instructions of the
//     FLY younger stages may stay in flight) and the consumers that their reads of stage g are complete (stage g + RING takes its slot).
//   * the 256-column form (NBW = 8: a consumer wave owns 64 pixels x 128 channels, 96 MFMAs per K-step) is the head towers' (Lambda_L2.py:44-51,
//     85-103): up to four convolutions of one geometry share the grid (aod_conv2d_grouped), the tile index names the group.
//
// Same products in the same order per accumulator as conv_igemm_kernel<.., X3 = true> (K-steps in (chunk, tap) order for C >= 256, (tap, chunk)
// otherwise; xh*wh, xl*wh, xh*wl per step) and the same fp32 epilogue arithmetic: IDENTICAL BITS (tests/test_gpu_x3p.py); only the fp32
// atomics of the optional column sums arrive in another order.
#include "conv_x3p.h"

namespace {

constexpr int XP_BM = 128, XP_XBYTES = 16384;
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
#ifdef AOD_TILE_TIMING
// debug build only (tools/dbg/x3p_timing.py): per-workgroup stamps of consumer wave 0 -- kernel entry, ring primed, K loop done, epilogue
// done (first tile), kernel exit -- shader clock (s_memtime) and 100 MHz wall clock
__device__ unsigned long long* g_x3p_stamps = nullptr;
extern "C" int aod_dbg_set_x3p_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x3p_stamps), &buf, sizeof(buf)); }
#define XSTAMP(k) do { if (g_x3p_stamps && threadIdx.x == 0) { g_x3p_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); g_x3p_stamps[(size_t)blockIdx.x * 16 + 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define XSTAMP(k) do {} while (0)
#endif
#define XP_GSEL(f) (gi == 0 ? p.grp[0].f : (gi == 1 ? p.grp[1].f : (gi == 2 ? p.grp[2].f : p.grp[3].f)))

// NBW: 16-channel blocks per consumer wave -- 4: 128-column tiles, 8: 256-column tiles.  LAT: X3PArgs.lat (TAPS = 4 = the most taps a class has)
// PRE (128-column tiles, dense destinations): the tile's epilogue operand -- 1: the residual rows (heads + tails, 64 registers), 2: the ReLU-mask
// rows (heads, 32 registers) -- is requested BEFORE the K loop and waited for after it.  For the short-K layers that carry one (the expand 1 x 1
// convs: 8 K-steps between 64 KB of residual reads and 64 KB of stores per tile; the dgrads of the reduce convs) the tile otherwise serialises
// K loop -> operand latency + transfer -> stores with nothing else on the CU to run meanwhile.
template <int TAPS, int NBW, int LAT = 0, int PRE = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_x3p_kernel(const X3PArgs p) {
  static_assert(LAT != 1 || TAPS == 4, "class-major form: up to four taps per class");
  static_assert(PRE == 0 || (NBW == 4 && LAT == 0), "operand prefetch: 128-column dense tiles");
  constexpr int BN = 32 * NBW, WBYTES = BN * 128, STAGE = XP_XBYTES + WBYTES;
  // ring: RING slots, FLY stages in flight.  Iteration k of a loader issues stage k, waits until all but the youngest FLY stages'
  // instructions have landed (stage k - FLY and older) and joins barrier k - FLY - 1, behind which the consumers read stage k - FLY;
  // stage k overwrites the slot of stage k - RING, whose reads ended before barrier k - RING <= k - FLY - 2: RING >= FLY + 2.
  constexpr int RING = NBW == 4 ? 5 : 3, FLY = RING - 2;
  constexpr int NL = 4 + NBW;                                   // LDS-DMA instructions per loader wave and stage
  static_assert(RING * STAGE <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tpc = p.tiles_m * p.tiles_n;                       // tiles per class (class-major form) = tiles per group otherwise
  const int tpg = tpc * (LAT == 1 ? 4 : 1);                    // tiles per group
  const int ntiles = tpg * p.ngroups;
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;
  const int nmine = (ntiles - bid + G - 1) / G;              // tiles bid, bid + G, ... (launcher: G <= ntiles)
  const int CC = p.C >> 6;                                    // 64-column (= 32-channel) chunks per tap
  auto tile_of = [&](int j) {
    const int base = j * G, left = ntiles - base;
    return base + xp_swizzle(bid, left < G ? left : G);
  };
  // K-steps of a tile: taps x chunks; in the class-major form the class (tile / tiles per class) decides the taps: 1, 2, 2, 4
  auto taps_of = [&](int tile) {
    if (LAT != 1) return TAPS;
    const int cls = (tile % tpg) / tpc;
    return ((cls >> 1) ? 2 : 1) * ((cls & 1) ? 2 : 1);
  };
  int total = 0;
  if (LAT == 1) { for (int j = 0; j < nmine; ++j) total += taps_of(tile_of(j)) * CC; }
  else total = nmine * TAPS * CC;
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
    const int kc = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);          // k-chunk this lane fetches: slot ^ ((row >> 1) & 7), the same for all its rows
    // filter rows: LDS row R of the tile holds output channel n0 + perm(R), perm = the order in which the transposed product leaves eight
    // consecutive channels in one lane (block pair a = 2 x 16 rows -> channels 32a + 8q + {0..3 | 4..7}, see the consumers' epilogue)
    unsigned wvo[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int R = 8 * w + 32 * i + prow;
      const int wv = R / (16 * NBW), nb = (R >> 4) % NBW;
      const int nrel = wv * (16 * NBW) + (nb >> 1) * 32 + ((R >> 2) & 3) * 8 + (nb & 1) * 4 + (R & 3);
      wvo[i] = (unsigned)((nrel * p.K + kc * 8) * 2);
    }
    const int sgn = p.transposed ? -1 : 1;
    int gi_ = 0, slot = 0;                                     // stages issued so far, ring slot of the next one
    char* const lds_w = smem + (8 * w) * 128;
    for (int j = 0; j < nmine; ++j) {
      const int tile = tile_of(j);
      const int gi = tile / tpg, tl0 = tile - gi * tpg;
      const int cls = LAT == 1 ? tl0 / tpc : 0, tl = tl0 - cls * tpc;
      const int py = cls >> 1, px = cls & 1;                      // class-major form: parity of the destination pixels of this tile
      const int tm = tl / p.tiles_n, tn = tl - tm * p.tiles_n;
      const int m0 = tm * XP_BM, n0 = tn * BN;
      // (p.rot bits 8 / 9, timing experiments only: a zero-record descriptor drops every load of that operand at the range check)
      const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)XP_GSEL(x), 0, (p.rot & 256) ? 0 : (int)p.x_bytes, 0x00020000);
      const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)XP_GSEL(w), 0, (p.rot & 512) ? 0 : (int)p.w_bytes, 0x00020000);
      // the tile lies inside ONE segment (launcher): workgroup-uniform geometry
      const int sg = LAT == 1 ? 0 : seg_of(m0);
      const unsigned mstart = sg ? (unsigned)xp_sel8(p.seg_mend, sg - 1) : 0u;
      const unsigned OW = (unsigned)xp_sel8(p.segOW, sg), OH = (unsigned)xp_sel8(p.segOH, sg);
      const unsigned H = (unsigned)xp_sel8(p.segH, sg), W = (unsigned)xp_sel8(p.segW, sg);
      // rows of the tile enumerate (image, y, x) over RH x RW: the destination map, or (class-major) the class's half-resolution lattice
      const unsigned RH = LAT == 1 ? OH >> 1 : OH, RW = LAT == 1 ? OW >> 1 : OW, rhw = RH * RW;
      const int mlim = LAT == 1 ? (int)(rhw * (unsigned)p.segB[0]) : p.M;
      const unsigned rowb = (unsigned)(p.C * 2);
      const unsigned src0b = (unsigned)((unsigned long long)xp_sel8(p.seg_src0, sg) * rowb);       // (32-bit byte offsets: x_bytes < 3.5 GiB)
      const unsigned imgb = H * W * rowb;
      const int ntap = taps_of(tile);
      unsigned vo[TAPS][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * w + 32 * i + prow;
        const bool live = m < mlim;
        const unsigned ml = (unsigned)m - mstart;
        const unsigned b = xp_udiv(ml, rhw), rem = ml - b * rhw, oy = xp_udiv(rem, RW), ox = rem - oy * RW;
        const unsigned base = src0b + b * imgb + (unsigned)(kc * 16);
        if constexpr (LAT == 1) {
          // destination pixel (2 oy + py, 2 ox + px) of a 3x3 / stride-2 / pad-1 conv's input gradient: tap (r, s) reaches it from dZ pixel
          // ((2 oy + py + 1 - r) / 2, (2 ox + px + 1 - s) / 2) when both are whole -- r = 1 for an even row, r in {0, 2} for an odd one -- i.e.
          // from dZ row oy + 1 (r = 0), oy (r = 1 or 2); likewise the columns.  Taps in ascending (r, s) order like the general kernel's
#pragma unroll
          for (int tq = 0; tq < 4; ++tq) {
            const int ri = px ? tq >> 1 : tq, si = px ? tq & 1 : 0;            // tap tq of the class = (ri-th row tap, si-th column tap)
            const int sy = (int)oy + ((py && ri == 0) ? 1 : 0), sx = (int)ox + ((px && si == 0) ? 1 : 0);
            const bool ok = live && tq < ntap && (unsigned)sy < H && (unsigned)sx < W;
            vo[tq][i] = ok ? base + (unsigned)(sy * (int)W + sx) * rowb : XP_OOB;
          }
        } else {
          const int y0 = p.transposed ? (int)oy + p.pad : (int)oy * p.stride - p.pad;
          const int x0 = p.transposed ? (int)ox + p.pad : (int)ox * p.stride - p.pad;
          constexpr int RR = TAPS == 9 ? 3 : 1;
#pragma unroll
          for (int r = 0; r < RR; ++r) {
            const int y = y0 + sgn * r * p.dil;
            const bool yok = live && (unsigned)y < H;
#pragma unroll
            for (int s_ = 0; s_ < RR; ++s_) {
              const int x = x0 + sgn * s_ * p.dil;
              const bool ok = yok && (unsigned)x < W;
              vo[r * RR + s_][i] = ok ? base + (unsigned)(y * (int)W + x) * rowb : XP_OOB;
            }
          }
        }
      }
      const unsigned soff_w0 = (unsigned)n0 * (unsigned)(p.K * 2);
      auto issue = [&](const unsigned (&v)[4], int t, int cc) {
        char* const sa = lds_w + slot * STAGE;
        const unsigned soff_x = (unsigned)cc * 128u;
        const unsigned soff_w = soff_w0 + (unsigned)((t * p.C + cc * 64) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned off = v[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sa + i * 4096), 16, off, soff_x, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          const unsigned off = wvo[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sa + XP_XBYTES + i * 4096), 16, off, soff_w, 0, 0);
        }
        ++gi_;
        slot = slot == RING - 1 ? 0 : slot + 1;
        if (gi_ > FLY) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * FLY) : "memory");
          __builtin_amdgcn_s_barrier();
        }
      };
      if constexpr (LAT == 1) {
        // (tap, chunk) order over the class's taps; filter tap index of tap tq: (r, s) = (py ? 2 ri : 1, px ? 2 si : 1)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
          if (tq < ntap) {
            const int ri = px ? tq >> 1 : tq, si = px ? tq & 1 : 0;
            const int tapidx = (py ? 2 * ri : 1) * 3 + (px ? 2 * si : 1);
            for (int cc = 0; cc < CC; ++cc) issue(vo[tq], tapidx, cc);
          }
        }
      } else if (TAPS == 1 || p.tapin) {
        // (p.rot low bits, timing experiments only: the chunk loop starts at a tile-dependent chunk -- another summation order)
        int cc = (p.rot & 255) ? (tm * (p.rot & 255)) % CC : 0;
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
    const int done = total > FLY ? total - FLY : 0;
    for (int k = done; k < total + 1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }

  // ========================================================================== consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int sw = (lr >> 1) & 7;
  const int oh = (lq ^ sw) << 4, ol = ((4 + lq) ^ sw) << 4;          // 16-B slot of this lane's head / tail k-chunk inside a 128-B tile row
  const int xoff = (wm * 64 + lr) * 128, woff = XP_XBYTES + (wn * 16 * NBW + lr) * 128;
  const int NP = ((p.N + 31) >> 5) << 6;                             // destination row pitch (X-layout, elements)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_scale, 0, p.pre_scale ? p.N * 4 : 0, 0x00020000);     // empty descriptor -> zeros

  XSTAMP(0);
  __builtin_amdgcn_s_barrier();                                      // stage 0 is resident
  XSTAMP(1);
  int cslot = 0;
  for (int j = 0; j < nmine; ++j) {
    const int tile = tile_of(j);
    const int gi = tile / tpg, tl0 = tile - gi * tpg;
    const int cls = LAT == 1 ? tl0 / tpc : 0, tl = tl0 - cls * tpc;
    const int tm = tl / p.tiles_n, tn = tl - tm * p.tiles_n;
    const int m0 = tm * XP_BM, n0 = tn * BN;
    const int nk = taps_of(tile) * CC;
    f32x4 acc[NBW][4];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // destination rows of this lane's four pixels (used by the epilogue; PRE: already here, for the operand requests)
    bool live[4];
    long long prow_[4];
    auto row_addresses = [&]() {
      const int sg = LAT == 1 ? 0 : seg_of(m0);
      const long long drow0 = xp_sel8(p.seg_dst0, sg) - (long long)(sg ? xp_sel8(p.seg_mend, sg - 1) : 0);
      const unsigned eOW = (unsigned)p.segOW[0], eOH = (unsigned)p.segOH[0];
      const unsigned eRW = LAT == 1 ? eOW >> 1 : eOW, erhw = (LAT == 1 ? eOH >> 1 : eOH) * eRW;
      const int mlim = LAT == 1 ? (int)(erhw * (unsigned)p.segB[0]) : p.M;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int m = m0 + wm * 64 + mb * 16 + lr;
        live[mb] = m < mlim;
        // (rows past the end read the last row's operands instead of branching per lane around the loads: a divergent branch per load makes
        // the compiler wait for each one where the paths merge; only the stores are predicated)
        const int mc = live[mb] ? m : mlim - 1;
        if constexpr (LAT == 0) prow_[mb] = (drow0 + mc) * NP;
        else {
          // lattice destinations: row (image, y, x) of the tile's row space lands at (2y + py, 2x + px) of the full-resolution map
          const unsigned b = xp_udiv((unsigned)mc, erhw), rem = (unsigned)mc - b * erhw, oy = xp_udiv(rem, eRW), ox = rem - oy * eRW;
          const long long drow = LAT == 1 ? ((long long)b * eOH + 2 * oy + (unsigned)(cls >> 1)) * eOW + 2 * ox + (unsigned)(cls & 1)
                                          : (long long)b * p.up_hw + (long long)(2 * oy) * p.up_w + 2 * ox;
          prow_[mb] = (xp_sel8(p.seg_dst0, 0) + drow) * NP;
        }
      }
    };
    bf16x8 prh[PRE ? 2 : 1][4], prl[PRE == 1 ? 2 : 1][4];
    if constexpr (PRE != 0) {
      row_addresses();
      const bf16_t* const src = PRE == 1 ? p.res : XP_GSEL(mask);
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const int col = 2 * (n0 + wn * 16 * NBW + 32 * aa) + 8 * lq;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          prh[aa][mb] = *reinterpret_cast<const bf16x8*>(src + prow_[mb] + col);
          if constexpr (PRE == 1) prl[aa][mb] = *reinterpret_cast<const bf16x8*>(src + prow_[mb] + col + 32);
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // (the requests leave here, not where hipcc would sink them: in front of their first use)
    }
    for (int ks = 0; ks < nk; ++ks) {
      const char* const st = smem + cslot * STAGE;
      cslot = cslot == RING - 1 ? 0 : cslot + 1;
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xh[q] = *reinterpret_cast<const bf16x8*>(st + xoff + oh + q * 2048);
#pragma unroll
      for (int h = 0; h < NBW / 4; ++h) {
        bf16x8 wh[4], wl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wh[q] = *reinterpret_cast<const bf16x8*>(st + woff + oh + (h * 4 + q) * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[h * 4 + q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], xh[mb], acc[h * 4 + q][mb], 0, 0, 0);
        if (h == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xl[q] = *reinterpret_cast<const bf16x8*>(st + xoff + ol + q * 2048);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[h * 4 + q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], xl[mb], acc[h * 4 + q][mb], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) wl[q] = *reinterpret_cast<const bf16x8*>(st + woff + ol + (h * 4 + q) * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[h * 4 + q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[q], xh[mb], acc[h * 4 + q][mb], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    if (j == 0) XSTAMP(2);
    if (j == 1) XSTAMP(5);
    // ---- epilogue, in registers.  acc[2a][mb][r] = channel cb + r, acc[2a + 1][mb][r] = channel cb + 4 + r of pixel m0 + wm * 64 + mb * 16 + lr,
    // cb = n0 + wn * 16 NBW + 32a + 8 * lq: head columns 2 * (cb - 8 lq) + 8 lq .. + 7 of the destination row, tails 32 columns further
    bf16_t* const gy = XP_GSEL(y);
    const bf16_t* const gmask = XP_GSEL(mask);
    float* const gcs = XP_GSEL(colsum);
    const float* const gshift = XP_GSEL(shift);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)gshift, 0, gshift ? p.N * 4 : 0, 0x00020000);
    if constexpr (PRE == 0) row_addresses();
    // PG block pairs (32 channels each) at a time: all their operands are requested before the first is used.  Two pairs with 64 accumulator
    // registers; one with 128 (the 256-column tile), where a second pair's 48 operand registers would spill
    constexpr int PG = NBW == 4 ? 2 : 1;
#pragma unroll
    for (int pg = 0; pg < NBW / 2 / PG; ++pg) {
      float cs1[PG][8], cb1[PG][8], csum[PG][8];
      bf16x8 rh[PG][4], rl[PG][4], mh[PG][4];
#pragma unroll
      for (int aa = 0; aa < PG; ++aa) {
        const int cb = n0 + wn * 16 * NBW + 32 * (pg * PG + aa) + 8 * lq;
        const int col = 2 * (cb - 8 * lq) + 8 * lq;
        const u32x4_t b0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, cb * 4, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, cb * 4 + 16, 0, 0);
        // (the scale vector only where there is one: a load whose result is never used is never waited for either, and the registers it
        // targets stay 'pending' into the next tile's K loop -- whose first fragment reads then wait for EVERY older memory operation of the
        // wave, i.e. for the previous tile's stores to drain)
        u32x4_t s0 = {0u, 0u, 0u, 0u}, s1 = {0u, 0u, 0u, 0u};
        if (p.pre_scale) { s0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, cb * 4, 0, 0); s1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, cb * 4 + 16, 0, 0); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cb1[aa][q] = __uint_as_float(b0[q]); cb1[aa][4 + q] = __uint_as_float(b1[q]);
          cs1[aa][q] = __uint_as_float(s0[q]); cs1[aa][4 + q] = __uint_as_float(s1[q]);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          if constexpr (PRE == 1) { rh[aa][mb] = prh[aa][mb]; rl[aa][mb] = prl[aa][mb]; }
          else if constexpr (PRE == 2) mh[aa][mb] = prh[aa][mb];
          else {
            if (p.res) {
              rh[aa][mb] = *reinterpret_cast<const bf16x8*>(p.res + prow_[mb] + col);
              rl[aa][mb] = *reinterpret_cast<const bf16x8*>(p.res + prow_[mb] + col + 32);
            }
            if (gmask) mh[aa][mb] = *reinterpret_cast<const bf16x8*>(gmask + prow_[mb] + col);
          }
        }
      }
#pragma unroll
      for (int aa = 0; aa < PG; ++aa) {
        const int a = pg * PG + aa;
        const int col = 2 * (n0 + wn * 16 * NBW + 32 * a) + 8 * lq;
#pragma unroll
        for (int q = 0; q < 8; ++q) csum[aa][q] = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          // (rows past the end are COMPUTED like the others, on the last row's operands, and only their stores / column sums are predicated:
          // with the uses behind a per-lane branch hipcc's wait-count pass carries the operand loads as 'possibly pending' across the join
          // into the next tile's K loop, whose first fragment read then waited vmcnt(0) -- i.e. for this tile's STORES to drain)
          float v[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[q] = acc[2 * a][mb][q]; v[4 + q] = acc[2 * a + 1][mb][q]; }
          if (p.pre_scale) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= cs1[aa][q];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += cb1[aa][q];
          if (p.res) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rh[aa][mb][q] + (float)rl[aa][mb][q];
          }
          if (gmask) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ((float)mh[aa][mb][q] > 0.f) ? v[q] : 0.f;
          }
          if (p.relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          bf16x8 ov, ovl;
#pragma unroll
          for (int q = 0; q < 8; ++q) { csum[aa][q] += live[mb] ? v[q] : 0.f; ov[q] = (bf16_t)v[q]; }
#pragma unroll
          for (int q = 0; q < 8; ++q) ovl[q] = (bf16_t)(v[q] - (float)ov[q]);
          if (live[mb]) {
            *reinterpret_cast<bf16x8*>(gy + prow_[mb] + col) = ov;
            *reinterpret_cast<bf16x8*>(gy + prow_[mb] + col + 32) = ovl;
          }
        }
      }
      if (gcs) {
        // column sums of these 32 PG channels: the 16 lanes of a quarter (same lq) hold the same 8 PG channels -- butterfly over lr, then lane lr
        // sends channel (lr >> 3, lr & 7) of its quarter: ONE atomic wave-instruction with distinct, contiguous words per wave and pair group
        // (four-lane atomics per channel -- 64 instructions per tile into the same 256 words from every CU -- ran the layer-3 dgrad 28 %
        // slower than the general kernel, whose tile sends two full-wave atomics)
        float mine = 0.f;
#pragma unroll
        for (int aa = 0; aa < PG; ++aa)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float s = csum[aa][q];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            mine = lr == aa * 8 + q ? s : mine;
          }
        if (lr < 8 * PG) atomicAdd(gcs + n0 + wn * 16 * NBW + 32 * (pg * PG + (lr >> 3)) + 8 * lq + (lr & 7), mine);
      }
    }
    if (j == 0) XSTAMP(3);
    if (j == 1) XSTAMP(6);
  }
  XSTAMP(4);
}
#undef XP_GSEL

int g_x3p_cus[64];
long long g_x3p_count = 0;

template <int TAPS, int NBW, int LAT = 0, int PRE = 0>
int xp_launch(const X3PArgs& a, int grid, hipStream_t st) {
  constexpr int STAGE = XP_XBYTES + 32 * NBW * 128, RING = NBW == 4 ? 5 : 3;
  const size_t lds = (size_t)RING * STAGE;
  static unsigned long long attr = 0;
  if (aod_first_on_device(&attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<TAPS, NBW, LAT, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_x3p_kernel<TAPS, NBW, LAT, PRE>), dim3(grid), dim3(512), lds, st, a);
  return 0;
}

}  // namespace

// GEMM rows a launch tiles: the destination pixels, or (class-major form) the pixels of ONE of the four classes
static long long xp_rows(const X3PArgs& a) {
  return a.lat == 1 ? (long long)a.segB[0] * (a.segOH[0] / 2) * (a.segOW[0] / 2) : a.M;
}

// Operand-prefetch form (conv_x3p_kernel PRE): dense 1x1 launches with EXACTLY one of residual (-> 1) / ReLU mask (-> 2), else 0.
// AOD_X3P_PRE=0 disables.
static int xp_pre(const X3PArgs& a) {
  const char* e = getenv("AOD_X3P_PRE");
  if (e && e[0] == '0') return 0;
  if (a.taps != 1 || a.lat != 0 || a.ngroups > 1 || a.stride != 1) return 0;
  const bf16_t* mask = a.ngroups == 1 ? a.grp[0].mask : a.mask;
  if ((a.res != nullptr) == (mask != nullptr)) return 0;
  return a.res ? 1 : 2;
}

// 256-column tiles (half the pixel bytes per MFMA, twice the MFMAs per barrier) when they still give at least three quarters of the CUs a
// tile; AOD_X3P_BN=128 / 256 forces
static bool xp_wide(const X3PArgs& a) {
  if (xp_pre(a)) return false;                  // (the prefetch form exists for the 128-column tile: 64 accumulators leave room for the operand)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  if (!g_x3p_cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_x3p_cus[dev] = n;
  }
  if (a.N % 256 != 0) return false;
  const char* fbn = getenv("AOD_X3P_BN");
  if (fbn) return atoi(fbn) == 256;
  const int ng = a.ngroups > 1 ? a.ngroups : 1;
  return ((xp_rows(a) + XP_BM - 1) / XP_BM) * (a.N / 256) * ng * (a.lat == 1 ? 4 : 1) >= (long long)g_x3p_cus[dev] * 3 / 4;
}

int aod_conv_x3p_wants(const X3PArgs& a, int deterministic_colsum) {
  const char* now = getenv("AOD_X3P");                 // (read per call: tests switch it in-process)
  if (now && now[0] == '0') return 0;
  if (a.taps != 1 && a.taps != 9) return 0;
  if (a.taps == 9 && a.S != 3) return 0;
  if (a.N % 128 != 0 || a.C % 64 != 0 || a.M <= 0) return 0;
  // AOD_X3P_DGRAD=0 (parallel.GradSync sets it when gradients are all-reduced under the backward pass): backward launches stay with the general
  // kernel.  A persistent grid of one 160-KB-LDS workgroup per CU assumes every CU is free; RCCL's channel workgroups hold some for milliseconds,
  // and the workgroups that cannot start then run AFTER the others -- a one-round launch takes twice as long -- where the general kernel's many
  // small workgroups just lose those CUs' share.  Forward and scoring launches never run beside a collective.
  if (a.transposed) { const char* dg = getenv("AOD_X3P_DGRAD"); if (dg && dg[0] == '0') return 0; }
  if (a.lat == 1) {
    // class-major stride-2 dgrad: 3x3, pad 1, no dilation, one segment with an even map whose halves are the dZ map
    const char* lt = getenv("AOD_X3P_LATTICE");
    if (lt && lt[0] == '0') return 0;
    if (!(a.transposed && a.stride == 2 && a.taps == 9 && a.pad == 1 && a.dil == 1 && a.nseg == 1 && a.ngroups <= 1)) return 0;
    if ((a.segOH[0] & 1) || (a.segOW[0] & 1) || a.segH[0] * 2 != a.segOH[0] || a.segW[0] * 2 != a.segOW[0]) return 0;
  } else if (a.lat == 2) {
    const char* lt = getenv("AOD_X3P_LATTICE");
    if (lt && lt[0] == '0') return 0;
    if (!(a.transposed && a.stride == 1 && a.taps == 1 && a.pad == 0 && a.nseg == 1 && a.ngroups <= 1 && a.up_w > 0)) return 0;
  } else if (a.transposed && a.stride != 1) return 0;
  if (deterministic_colsum) return 0;                  // ordered column sums stay with the general kernel (determinism.hip)
  if ((long long)a.K * 2 * a.N >= 0x7fffffffll) return 0;
  // a tile needs a K loop long enough to amortise the one-workgroup-per-CU structure (its epilogue runs beside nothing but the loaders' next
  // stages): from ~24 K-steps of a 128-column tile on the persistent form wins -- every 3x3 layer (36+ steps), the 1024 -> 256 reduce / lateral
  // 1x1 convs (32 steps: 37.0 -> 28.7 us), retina_cls' dgrad (54: 243 -> 230 us) -- and a 256-column tile's K-step carries twice the matrix
  // work per barrier, so 12 of those do (the 512 -> 256 lateral and the 512 -> 1024 stride-2 downsample conv: 57.7 -> 55.0, 57.0 -> 54.3 us).
  // Below that the general kernel's two workgroups per CU win (the expand 1x1 convs with their residual, 4 - 16 steps: 45.5 vs 49.9,
  // 34.0 vs 36.8 us; retina_reg / retina_L dgrads with 18 / 9 steps of a 128-column tile: 106 vs 116, 76 vs 88 us) -- a residual epilogue is
  // only taken from 24 steps on.  The class-major form is judged by its average class (9 taps over 4 classes).
  // profiles/r06_x3p_micro.txt; AOD_X3P_MIN_STEPS overrides the threshold.
  {
    const char* ms = getenv("AOD_X3P_MIN_STEPS");
    const int thr = ms ? atoi(ms) : 24;
    const int steps = a.lat == 1 ? 9 * (a.C >> 6) / 4 : a.taps * (a.C >> 6);
    // (the operand-prefetch form, xp_pre, does not change this: with the residual requested before the K loop the 8-step expand conv of
    // layer 3 still takes 49.5 us against the general kernel's 44.9 -- its K loop is bound by what 96 KB of ring can keep in flight against
    // the memory latency under the residual / store traffic, 1.08 us per K-step by the stamps -- and only the 4-step layer-2 expand conv wins,
    // 73.8 against 80.0 us; AOD_X3P_PRE_MIN_STEPS lowers the threshold for launches that qualify for the form)
    const char* pms = getenv("AOD_X3P_PRE_MIN_STEPS");
    if (pms && xp_pre(a)) { if (steps < atoi(pms)) return 0; }
    else
    if (a.lat != 1 && (steps * (xp_wide(a) ? 2 : 1) < thr || (a.res && steps < thr))) return 0;
    if (a.lat == 1 && steps * (xp_wide(a) ? 2 : 1) < thr / 2) return 0;          // (the general kernel's class-major launches are its slowest: 86 - 127 TFLOP/s)
  }
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
  const int ng = a.ngroups > 1 ? a.ngroups : 1;
  const long long tiles = ((xp_rows(a) + XP_BM - 1) / XP_BM) * (a.N / 128) * ng * (a.lat == 1 ? 4 : 1);
  if (tiles < (mint ? atoll(mint) : 192)) return 0;
  return 1;
}

int aod_conv_x3p_launch(const X3PArgs& a0, hipStream_t st) {
  X3PArgs a = a0;
  if (a.ngroups < 1) {
    a.ngroups = 1;
    a.grp[0].x = a.x; a.grp[0].w = a.w; a.grp[0].y = a.y; a.grp[0].shift = a.pre_shift; a.grp[0].mask = a.mask; a.grp[0].colsum = a.colsum;
  }
  { const char* r = getenv("AOD_X3P_ROT"); a.rot = r ? atoi(r) : 0; }
  const bool wide = xp_wide(a);                 // (also resolves the CU count of the current device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  const int ncu = g_x3p_cus[dev] ? g_x3p_cus[dev] : 256;
  a.tiles_m = (int)((xp_rows(a) + XP_BM - 1) / XP_BM);
  a.tiles_n = a.N / (wide ? 256 : 128);
  const long long ntiles = (long long)a.tiles_m * a.tiles_n * a.ngroups * (a.lat == 1 ? 4 : 1);
  const int grid = (int)(ntiles < ncu ? ntiles : ncu);
  if (a.lat == 1) { if (wide) xp_launch<4, 8, 1>(a, grid, st); else xp_launch<4, 4, 1>(a, grid, st); }
  else if (a.lat == 2) { if (wide) xp_launch<1, 8, 2>(a, grid, st); else xp_launch<1, 4, 2>(a, grid, st); }
  else if (a.taps == 1) {
    const int pre = xp_pre(a);
    if (wide) xp_launch<1, 8>(a, grid, st);
    else if (pre == 1) xp_launch<1, 4, 0, 1>(a, grid, st);
    else if (pre == 2) xp_launch<1, 4, 0, 2>(a, grid, st);
    else xp_launch<1, 4>(a, grid, st);
  }
  else { if (wide) xp_launch<9, 8>(a, grid, st); else xp_launch<9, 4>(a, grid, st); }
  AOD_LAUNCH_CHECK();
  __atomic_add_fetch(&g_x3p_count, 1, __ATOMIC_RELAXED);
  return 0;
}

extern "C" int64_t aod_conv_x3p_count(void) { return (int64_t)__atomic_load_n(&g_x3p_count, __ATOMIC_RELAXED); }
