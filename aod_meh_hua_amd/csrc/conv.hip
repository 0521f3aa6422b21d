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
#include "pointwise.h"
#include "conv_x3p.h"

struct ConvGroup { const bf16_t* x; const bf16_t* w; void* y; const float* pre_shift; const bf16_t* mask; float* colsum; };
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
  float* colsum;      // optional fp32 [N]: += column sums of the stored values (fused bias / BN-shift gradient in dgrad launches)
  int C, N, K, R, S, stride, pad, dil, transposed, relu, out_f32;
  int nseg, M;
  int tiles_m, tiles_n;
  int ksplit;         // > 1: split-K launch -- K-steps are divided over ksplit workgroups per tile, each stores its fp32 partial tile
  float* ws;          // split-K workspace, fp32 [ksplit][M][N] in GEMM-row order; conv_splitk_finalize sums the slabs IN ORDER (deterministic)
  int perm;           // dgrad of a stride-2 conv: GEMM rows run CLASS-MAJOR inside a segment (the four (y & 1, x & 1) classes of the
                      // destination pixels one after the other), so a tile is one class and the filter taps that can never hit it are skipped
  int ngroups;        // > 1: grouped launch (aod_conv2d_grouped): grp[] replaces x / w / y / pre_shift / mask / colsum
  ConvGroup grp[4];
  int stagger;        // 8-wave forms: waves 4-7 run half a K-step behind waves 0-3 (see the K loop)
  int x3;             // reference-precision mode (aod_conv_desc_t.x3): operands in the X-layout, three MFMAs per 32 channels (see X3 below)
  int tap_inner;      // X3: K-steps run (channel chunk, tap) with the TAP innermost (see the loaders)
  int bigrows;        // some segment has >= 2^22 rows: the float-reciprocal row decode is not exact, use integer division
  float* cs_ws;       // deterministic mode (determinism.hip): partial column sums go to row (group * tiles_m + tile_m) of this scratch [..][N]
                      // (split-K finalize: row = its row block) instead of into fp32 atomics; the launcher adds the rows in order afterwards
  int up_w, up_hw;    // > 0: LATTICE launch (conv_params): the GEMM rows are the pixels (b, y, x) of the source map and row (b, y, x) is stored
                      // at row b * up_hw + 2y * up_w + 2x of the destination -- the in-place 1x1 / stride-2 dgrad, whose other rows do not change
  long long x_bytes, w_bytes;
  int segH[8], segW[8], segOH[8], segOW[8], segB[8];
  long long seg_src0[8], seg_dst0[8];
  int seg_mend[8];
};

#ifdef AOD_TILE_TIMING
// debug build only (tools/dbg/tile_timing.py): per-workgroup wall-clock stamps (100 MHz) at the phase boundaries of the tile
__device__ unsigned long long* g_tile_stamps = nullptr;
extern "C" int aod_dbg_set_tile_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &buf, sizeof(buf)); }
#define TSTAMP(k) do { if (g_tile_stamps && threadIdx.x == 0) g_tile_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TSTAMP(k) do {} while (0)
#endif

// n / d for n < 2^22 (the float product is within 1 of the quotient; one correction step makes it exact)
__device__ __forceinline__ unsigned udiv_small(unsigned n, unsigned d) {
  unsigned q = (unsigned)((float)n * __builtin_amdgcn_rcpf((float)d));
  const int r = (int)(n - q * d);
  q = r < 0 ? q - 1 : ((unsigned)r >= d ? q + 1 : q);
  return q;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // bijective remap: blocks that share an XCD (bid % 8) get a contiguous range of tiles
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// OPS: which optional epilogue operands the instance supports -- 0 none, 1 ReLU mask only, 2 residual and mask (their prefetch registers
// are what pushes the 8-wave form into spills, so it exists without them)
// X3: the REFERENCE-PRECISION form (fp32-equivalent products on the bf16 matrix pipe).  Every fp32 value v travels as a bf16 head and a
// bf16 tail, v ~= h + l (h = bf16(v), l = bf16(v - h): 16 significant bits), in the X-LAYOUT: a tensor of C logical channels has
// 2 * ceil32(C) bf16 columns, [h(0..31) | l(0..31) | h(32..63) | l(32..63) | ...].  A 64-column K-step of such an operand is then 32
// channels whose heads are k-block 0 and whose tails are k-block 1, for the activations and for the packed filters alike, and
//     x * w ~= xh * wh + xl * wh + xh * wl            (the dropped xl * wl is 2^-16 of the product)
// is THREE MFMAs on fragments the unmodified staging already put into the LDS (the plain form issues two per K-step): same tiles, same
// LDS-DMA gather, same barriers.  The epilogue computes in fp32 as before (BN / bias / residual = head + tail / mask / ReLU / column
// sums) and writes head and tail of each value, 64 B apart.  p.N stays the LOGICAL output channel count (vector operands, fp32
// destinations); p.C / p.K are the physical (X-layout) widths of the source and of the packed filter rows.
template <int BM, int BN, int NT, int OPS, bool GROUPED, int ST, bool X3>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(BM >= 192 ? 2 : (NT == 512 ? 4 : 1), BM >= 192 ? 2 : (NT == 512 ? 4 : 8)))) void conv_igemm_kernel(const ConvKParams p) {
  static_assert(NT == 256 || NT == 512, "4 or 8 waves");
  static_assert(ST == 2 || (ST == 3 && NT == 256), "LDS stages: 2, or 3 for the 4-wave forms");
  static_assert(!X3 || NT == 256 || (NT == 512 && (BM == 256 || BM == 192 || BM == 128)), "X3: the 4-wave forms, the 8-wave 256 x 256 / 192 x 192 tiles and (A/B) the 8-wave 128 x 128 tile");
  constexpr int BK = 64;
  constexpr int CPR = BK / 8;        // 16-B chunks per tile row
  constexpr int RPP = NT / CPR;      // tile rows covered per pass of the NT threads
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int ROWB = BK * 2;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int WAVES_M = NT / 128;        // waves are laid out WAVES_M x 2
  constexpr int WM = BM / WAVES_M, WN = BN / 2;  // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int CP = BN + 4;               // fp32 epilogue pitch
  // the fp32 epilogue image of a 256 x 256 tile (266 KB) does not fit the LDS: it is written and stored in EP passes of EBM rows
  constexpr int EP = (BM * CP * 4 > 144 * 1024) ? 2 : 1;
  constexpr int EBM = BM / EP;
  constexpr bool PREFETCH = EP == 1 && !X3;       // residual / mask rows prefetched into registers before the K loop (not for the big tile: 16 x 8 regs; not in X3: twice the pieces)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  TSTAMP(0);
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  // class-major launches keep the round-robin block -> XCD assignment: the classes differ in work (1 / 2 / 2 / 4 taps of a 3x3, 1 / 0 / 0 / 0
  // of a 1x1) and contiguous per-XCD tile ranges would hand whole classes to single XCDs
  int wg;
  if (!p.perm) wg = xcd_swizzle(blockIdx.x, nwg * (GROUPED ? p.ngroups : p.ksplit));
  else if ((nwg & 31) == 0) {
    // ... but inside each quarter of the tile range (~ one class) every XCD still takes a contiguous chunk (neighbouring column tiles
    // share their A rows in that XCD's L2), and the quarters are interleaved in launch order
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, q = j & 3, k = j >> 2, tq = nwg >> 2;
    wg = q * tq + x * (tq >> 3) + k;
  } else wg = blockIdx.x;
  const int tile = wg % nwg, zz = wg / nwg;
  const int kz = GROUPED ? 0 : zz;           // kz: which slice of the K-steps (split-K launches)
  // grouped launch: `ngroups` convolutions of identical geometry (the cls / reg / evidence towers at one depth) share one grid, so
  // that their tiles fill whole rounds of the CUs together; group = which operand set this workgroup uses
  const int gi = GROUPED ? zz : 0;
  // (constant indices only: a run-time index into the by-value argument struct would move the whole struct to scratch memory; the
  // selection exists in the GROUPED instances only -- in the 256 x 256 tile, which sits at the register cap, it costs ~15 %)
#define AOD_GSEL(f, dflt) (GROUPED ? (gi == 0 ? p.grp[0].f : (gi == 1 ? p.grp[1].f : (gi == 2 ? p.grp[2].f : p.grp[3].f))) : (dflt))
  const bf16_t* const g_x = AOD_GSEL(x, p.x);
  const bf16_t* const g_w = AOD_GSEL(w, p.w);
  const float* const g_shift = AOD_GSEL(pre_shift, p.pre_shift);
#define g_y AOD_GSEL(y, p.y)
#define g_mask AOD_GSEL(mask, p.mask)
#define g_colsum AOD_GSEL(colsum, p.colsum)
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
#ifdef AOD_TILE_TIMING
  if (m0 < 0) return;      // (forces the argument loads ahead of the stamp)
  TSTAMP(8);
#endif

  // ---- operand staging: LDS-DMA (buffer_load_dwordx4 ... lds), no VGPR round trip, no ds_write.
  // One wave-instruction fills 1 KiB = 8 tile rows x 8 chunks LINEARLY (lane l -> row l>>3, slot l&7); the
  // XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side:
  // slot s of row r holds k-chunk s ^ ((r>>1)&7).  Out-of-image taps / K,N tails use an out-of-range
  // buffer offset, for which the hardware range check returns zeros (no select, no predication).
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int prow = lane >> 3;                       // row inside the wave's 8-row group
  const int kc = (lane & 7) ^ ((4 * uw + (lane >> 4)) & 7);   // k-chunk this lane fetches (same for every pass)
  const int C8 = p.C >> 3;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)g_x, 0, (int)p.x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)g_w, 0, (int)p.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;

  unsigned rbase[A_IT];                             // byte offset of pixel (b, 0, 0) of the row's source block
  int ry0[A_IT], rx0[A_IT], rH[A_IT], rW[A_IT];
  // segment of a row: seg_mend[] is padded with M, so counting the boundaries <= m needs no loop.  Nearly every tile lies inside ONE
  // segment: its index is then workgroup-uniform and the geometry comes from scalar loads; only tiles that straddle a segment boundary
  // take the per-lane path, whose indexed loads from the argument buffer are a serial chain of vector-memory round trips
  auto seg_of = [&](int m) {
    int sg = 0;
    if (p.nseg > 1) {
#pragma unroll
      for (int q = 0; q < 7; ++q) sg += (m >= p.seg_mend[q]) ? 1 : 0;
    }
    return sg;
  };
  const int m_last = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
  const int sg_first = seg_of(m0);
  const bool one_seg = sg_first == seg_of(m_last);
  struct Geo { unsigned mstart, OW, OH, ohw, B, H, W, src0b, imgb; };
  auto load_geo = [&](int sg) {
    Geo g;
    g.mstart = sg ? (unsigned)p.seg_mend[sg - 1] : 0u;
    g.OW = (unsigned)p.segOW[sg]; g.OH = (unsigned)p.segOH[sg]; g.ohw = g.OH * g.OW; g.B = (unsigned)p.segB[sg];
    g.H = (unsigned)p.segH[sg]; g.W = (unsigned)p.segW[sg];
    // byte offsets fit 32 bits (x_bytes < 3.5 GiB is checked on the host), so the arithmetic may wrap on the way
    g.src0b = (unsigned)((unsigned long long)p.seg_src0[sg] * (unsigned)(p.C * 2));
    g.imgb = g.H * g.W * (unsigned)(p.C * 2);
    return g;
  };
  const Geo gu = load_geo(sg_first);          // workgroup-uniform: scalar loads, once
  auto udiv = [&](unsigned n, unsigned d) { return p.bigrows ? n / d : udiv_small(n, d); };
  // (image, y, x) of the destination pixel with index ml inside its segment; `cls` = its (y & 1) * 2 + (x & 1) class in a class-major
  // launch (p.perm), where the segment's rows are ordered class 0 | class 1 | class 2 | class 3, each class image-major
  auto rowpos = [&](unsigned ml, const Geo& g, unsigned& b, unsigned& oy, unsigned& ox, int& cls) {
    if (!p.perm) {
      b = udiv(ml, g.ohw);
      const unsigned rem = ml - b * g.ohw;
      oy = udiv(rem, g.OW); ox = rem - oy * g.OW; cls = 0;
    } else {
      const unsigned H0 = (g.OH + 1) >> 1, H1 = g.OH >> 1, W0 = (g.OW + 1) >> 1, W1 = g.OW >> 1;
      const unsigned e0 = g.B * H0 * W0, e1 = e0 + g.B * H0 * W1, e2 = e1 + g.B * H1 * W0;
      cls = (ml >= e0 ? 1 : 0) + (ml >= e1 ? 1 : 0) + (ml >= e2 ? 1 : 0);
      const unsigned start = cls == 0 ? 0u : (cls == 1 ? e0 : (cls == 2 ? e1 : e2));
      const unsigned Hc = (cls & 2) ? H1 : H0, Wc = (cls & 1) ? W1 : W0, hw = Hc * Wc;
      const unsigned r = ml - start;
      b = udiv(r, hw);
      const unsigned rem = r - b * hw, yy = udiv(rem, Wc);
      oy = 2 * yy + (unsigned)(cls >> 1); ox = 2 * (rem - yy * Wc) + (unsigned)(cls & 1);
    }
  };
  const bool linear = one_seg && !p.perm && !p.up_w;       // destination rows of the tile are consecutive: drow = m + const
  // destination row of every tile row, parked in LDS behind the staging / epilogue area (read by the general epilogue, and by the
  // fast one on tiles that straddle a segment boundary)
  constexpr int EPI_BYTES = EBM * CP * 4;
  constexpr int DROW_OFF = (ST * STAGE > EPI_BYTES ? ST * STAGE : EPI_BYTES);
  long long* s_drow = reinterpret_cast<long long*>(smem + DROW_OFF);
  if (t < BM) {
    const int m = m0 + t;
    if (linear) s_drow[t] = p.seg_dst0[sg_first] + (long long)((unsigned)m - gu.mstart);
    else if (p.up_w) {          // (one segment: conv_params)
      unsigned b, oy, ox; int cls;
      rowpos((unsigned)m, gu, b, oy, ox, cls);
      s_drow[t] = p.seg_dst0[0] + (long long)b * p.up_hw + (long long)(2 * oy) * p.up_w + 2 * ox;
    }
    else if (!p.perm) { const int sg = m < p.M ? seg_of(m) : 0; s_drow[t] = p.seg_dst0[sg] + (m - (sg ? p.seg_mend[sg - 1] : 0)); }
    else {
      const int sg = one_seg ? sg_first : (m < p.M ? seg_of(m) : 0);
      const Geo g = one_seg ? gu : load_geo(sg);
      unsigned b, oy, ox; int cls;
      rowpos((unsigned)m - g.mstart, g, b, oy, ox, cls);
      s_drow[t] = p.seg_dst0[sg] + (long long)(b * g.ohw + oy * g.OW + ox);
    }
  }
  TSTAMP(10);
  // tap state of this lane's k-chunk
  const int nk_all = (p.K + BK - 1) / BK;
  const int kt_begin = (int)((long long)kz * nk_all / p.ksplit), kt_end = (int)((long long)(kz + 1) * nk_all / p.ksplit);
  int c8 = kc + kt_begin * CPR, tr = 0, ts = 0;
  if (c8 >= C8) { const int taps = c8 / C8; c8 -= taps * C8; tr = taps / p.S; ts = taps - tr * p.S; }

  unsigned wbase[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + RPP * i + 8 * uw + prow;
    wbase[i] = n < p.N ? (unsigned)(((long long)n * p.K + kc * 8 + (long long)kt_begin * BK) * 2) : OOB;
  }

  // A-operand byte offsets are kept incrementally: inside one filter tap consecutive K-steps only advance the
  // channel offset by BK elements (and the validity of the tap does not change), so the full coordinate / bounds
  // computation runs once per tap instead of once per K-step.  Invalid rows park at OOB_BASE, which stays out of
  // range under the increments (x_bytes < OOB_BASE is checked on the host).
  constexpr unsigned OOB_BASE = 0xf0000000u;
  const bool fast_tap = (C8 % CPR) == 0;          // then every lane of the block changes tap at the same K-step
  unsigned aoff[A_IT];
  unsigned woff[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) aoff[i] = OOB_BASE;
#pragma unroll
  for (int i = 0; i < B_IT; ++i) woff[i] = wbase[i] == OOB ? OOB_BASE : wbase[i];
  const int steps_per_tap = fast_tap ? C8 / CPR : 1;
  // taps that can reach this tile at all (class-major dgrad of a stride-2 conv: tap (r, s) only hits destination pixels with
  // y = r * dil - pad and x = s * dil - pad modulo 2); the others are skipped whole -- 5 to 8 of the 9 taps of a 3x3, 3 of 4 tiles
  // of a 1x1.  Rows are still checked one by one, so a tile that straddles two classes simply keeps every tap.
  unsigned long long tapmask = ~0ull;
  if (p.perm && fast_tap && one_seg) {
    unsigned b_, y_, x_; int c_first, c_last;
    rowpos((unsigned)m0 - gu.mstart, gu, b_, y_, x_, c_first);
    rowpos((unsigned)m_last - gu.mstart, gu, b_, y_, x_, c_last);
    if (c_first == c_last) {
      tapmask = 0ull;
      for (int r = 0; r < p.R; ++r)
        for (int q = 0; q < p.S; ++q)
          if ((((c_first >> 1) + p.pad - r * p.dil) & 1) == 0 && (((c_first & 1) + p.pad - q * p.dil) & 1) == 0) tapmask |= 1ull << (r * p.S + q);
    }
  }
  const bool skipping = tapmask != ~0ull;
  int kt_ld = kt_begin;                          // K-step the loaders fetch next
  // X3, TAPS INNERMOST.  In (tap, channel chunk) order a tile re-reads its pixel rows -- shifted by a pixel or a line -- once per tap,
  // C / 64 K-steps apart: 8 steps x 32 KB x 32 workgroups per XCD is twice the 4 MB L2, so every tap's re-read went out to the fabric (PMC:
  // 2 193 MB per grouped tower launch for 268 MB of activations, 8.2x).  In (channel chunk, tap) order the nine taps of a chunk follow each
  // other: the re-use distance is one K-step and the shifted rows are L2 / L1 hits.  Every K-step then changes the tap, so the incremental
  // offsets of the plain order do not apply: a row keeps its offset at tap (0, 0) and a validity bit per tap, a K-step adds the tap's
  // (workgroup-uniform per segment) displacement.  All x3 instances take this order (stride-1 forward / dgrad and strided forward; the
  // class-major stride-2 dgrad keeps the plain order with its tap skipping), so grouped and separate launches still agree bit for bit.
  const int ntaps = p.R * p.S;
  // (deep layers only: with C < 256 columns a tap is at most 3 K-steps long -- the re-reads are L2 hits already -- and the per-step offset
  // arithmetic shows: the 16-tap stem ran 13 % slower in this order)
  const bool tapin = X3 && p.tap_inner && !p.perm && !(p.transposed && p.stride != 1) && ntaps > 1 && ntaps <= 32 && p.C >= 256;
  int tp_tap = X3 ? kt_begin % ntaps : 0, tp_cc = X3 ? kt_begin / ntaps : 0;
  unsigned tbase[X3 ? A_IT : 1], tvm[X3 ? A_IT : 1], wb0[X3 ? B_IT : 1];
  int ksteps_in_tap = fast_tap ? kt_begin % steps_per_tap : 0;     // (a split-K slice may start inside a tap)
  bool new_tap = true;

  auto load_a = [&](int buf) {
    char* sa = smem + buf * STAGE;
    if constexpr (X3) {
      if (tapin) {
        const int trr = tp_tap / p.S, tss = tp_tap - trr * p.S;
        const int dy = trr * p.dil, dx = tss * p.dil;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int d = (dy * rW[i] + dx) * p.C * 2;
          const unsigned off = ((tvm[i] >> tp_tap) & 1u) ? tbase[i] + (unsigned)(p.transposed ? -d : d) + (unsigned)tp_cc * (BK * 2) : OOB_BASE;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sa + (RPP * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
        }
        return;
      }
    }
    if (new_tap) {
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
        aoff[i] = ok ? rbase[i] + (unsigned)(((y * rW[i] + x) * p.C + c8 * 8) * 2) : OOB_BASE;
      }
      new_tap = false;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const unsigned off = aoff[i];     // (a plain local: a subscript of a template-sized array is type-dependent and the host pass rejects it as a builtin argument)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sa + (RPP * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
    }
  };
  auto load_b = [&](int buf) {
    char* sb = smem + buf * STAGE + A_BYTES;
    if constexpr (X3) {
      if (tapin) {
        const unsigned kofs = (unsigned)((tp_tap * p.C + tp_cc * BK) * 2);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
          const unsigned off = wb0[i] == OOB ? OOB_BASE : wb0[i] + kofs;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sb + (RPP * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
        }
        return;
      }
    }
    const bool kok = (kt_ld * BK + kc * 8) < p.K;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const unsigned off = kok ? woff[i] : OOB_BASE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sb + (RPP * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
    }
  };
  auto skip_dead_taps = [&]() {      // (at a tap boundary) hop over the taps that cannot reach this tile
    while (tr < p.R && !((tapmask >> (tr * p.S + ts)) & 1ull)) {
      if (++ts == p.S) { ts = 0; ++tr; }
      kt_ld += steps_per_tap;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) woff[i] += (unsigned)(p.C * 2);
    }
  };
  auto advance = [&]() {      // state of the next K-step
    if constexpr (X3) {
      if (tapin) {
        ++kt_ld;
        if (++tp_tap == ntaps) { tp_tap = 0; ++tp_cc; }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) woff[i] += BK * 2;
    c8 += CPR;
    ++kt_ld;
    if (fast_tap && ++ksteps_in_tap < steps_per_tap) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) aoff[i] += BK * 2;
    } else {
      ksteps_in_tap = 0;
      while (c8 >= C8) { c8 -= C8; if (++ts == p.S) { ts = 0; ++tr; } }
      new_tap = true;
      if (skipping) skip_dead_taps();
    }
  };
  auto gload = [&](int buf) { load_a(buf); load_b(buf); advance(); };
  if (skipping) skip_dead_taps();

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  TSTAMP(1);
  // epilogue operands (bias / scale vectors of this thread's 8 columns; residual and ReLU mask of its E_IT row segments): issued
  // together with the first operand tile and consumed after the main loop -- a load-use chain per store iteration would expose one
  // memory latency per 16 B and cap memory-bound layers at a third of the bandwidth
  constexpr int NCH = BN / 8;              // 8-column chunks per tile row
  // (thread -> chunk mapping over a power-of-two chunk count: the 192-column tile leaves a quarter of its epilogue lanes idle)
  constexpr int NCHT = (NCH & (NCH - 1)) == 0 ? NCH : (NCH <= 32 ? 32 : 64);
  static_assert((EBM * NCHT) % NT == 0 && NT % NCHT == 0, "epilogue mapping");
  constexpr int E_IT = EBM * NCHT / NT;    // row segments per thread and epilogue pass
  const int ec = t % NCHT, er = t / NCHT;
  const bool ecok = NCHT == NCH || ec < NCH;
  float cs1[8], cb1[8], cs2[8];
  bf16x8 pres[PREFETCH ? E_IT : 1], pmask[PREFETCH ? E_IT : 1];
  // LATE (the 4-wave X3 forms, which cannot spare the registers during the K loop): the same operands are fetched once the accumulators
  // sit in LDS, all E_IT row segments at once, ahead of the barrier.  Fetched inside the store loop, every iteration's load waited behind the
  // previous iteration's store (the two may alias as far as the compiler knows): one exposed memory latency per 16-B segment, 9.4 us of
  // a 22 us tile on the residual convs of the X3 backbone.  (Not the 256 x 256 tile: its second-pass accumulators are still live, it spills.)
  constexpr bool LATE = X3 && EP == 1;
  bf16x8 lres[LATE ? E_IT : 1], lrl[(LATE && X3) ? E_IT : 1], lmask[LATE ? E_IT : 1];
  // interior tiles of the common configuration (bf16 destination, N % 8 == 0, no post-scale / raw copy) take an epilogue without per-thread
  // predicates; inside one segment the destination rows are also linear in m (drow = m + drow_lin)
  const bool fast = !p.out_f32 && !p.zraw && !p.post_scale && (p.N & 7) == 0 && n0 + BN <= p.N && m0 + BM <= p.M;
  // destination row pitch and column of logical channel n (elements): X-layout for the bf16 destinations of an X3 launch
  const bool xout = X3 && !p.out_f32;
  const int NP = xout ? ((p.N + 31) >> 5) << 6 : p.N;
  auto xcol = [&](int n) { return xout ? ((n >> 5) << 6) + (n & 31) : n; };
  const long long drow_lin = linear ? p.seg_dst0[sg_first] - (long long)gu.mstart : 0;
  const long long lin_off = (drow_lin + m0 + er) * NP + xcol(n0 + ec * 8);     // element offset of this thread's first row segment
  const long long lin_step = (long long)(NT / NCHT) * NP;                // ... and the distance to its next one
  auto prefetch_epilogue = [&](bool from_table) {
    const int n = n0 + ec * 8;
    if (fast) {
      {
        // bias through a buffer descriptor that is EMPTY when there is no bias: the range check then returns zeros and the load needs
        // no branch (a branch would make the compiler wait for the loaded values where the two paths merge, i.e. right here)
        const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)g_shift, 0, g_shift ? p.N * 4 : 0, 0x00020000);
        const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_scale, 0, p.pre_scale ? p.N * 4 : 0, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, n * 4, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, n * 4 + 16, 0, 0);
        const u32x4 s0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, n * 4, 0, 0), s1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, n * 4 + 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cb1[j] = __uint_as_float(b0[j]); cb1[4 + j] = __uint_as_float(b1[j]);
          cs1[j] = __uint_as_float(s0[j]); cs1[4 + j] = __uint_as_float(s1[j]);       // (zeros without a scale vector: not used then)
        }
      }
      if (!from_table) {
        if (PREFETCH && OPS > 1 && p.res) {
#pragma unroll
          for (int it = 0; it < E_IT; ++it) pres[it] = *reinterpret_cast<const bf16x8*>(p.res + lin_off + it * lin_step);
        }
        if (PREFETCH && OPS > 0 && g_mask) {
#pragma unroll
          for (int it = 0; it < E_IT; ++it) pmask[it] = *reinterpret_cast<const bf16x8*>(g_mask + lin_off + it * lin_step);
        }
        return;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = n + j < p.N;
        cs1[j] = (p.pre_scale && ok) ? p.pre_scale[n + j] : 1.f;
        cb1[j] = (g_shift && ok) ? g_shift[n + j] : 0.f;
        cs2[j] = (p.post_scale && ok) ? p.post_scale[n + j] : 1.f;
      }
    }
    if (PREFETCH && OPS > 0 && (p.res || g_mask)) {
#pragma unroll
      for (int it = 0; it < E_IT; ++it) {
        const int row = er + it * (NT / NCHT);
        const bool ok = (m0 + row < p.M) && (n + 8 <= p.N);
        const long long drow = from_table ? s_drow[row] : drow_lin + m0 + row;
        const long long off = ok ? drow * NP + xcol(n) : 0;
        if (OPS > 1 && p.res && ok) pres[it] = *reinterpret_cast<const bf16x8*>(p.res + off);
        if (OPS > 0 && g_mask && ok) pmask[it] = *reinterpret_cast<const bf16x8*>(g_mask + off);
      }
    }
  };
  // first stage: the weight tile and the epilogue operands do not depend on the row decode -- they go out first and are in flight
  // while the rows are resolved
  bool have = __builtin_amdgcn_readfirstlane(kt_ld) < kt_end;     // (kt_ld is the same in every lane; say so, or the loop control goes through EXEC)                    // (a class without a reachable tap has no K-step at all: its dX is the epilogue of zero)
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int n = n0 + RPP * i + 8 * uw + prow;
      wb0[i] = n < p.N ? (unsigned)(((long long)n * p.K + kc * 8) * 2) : OOB;
    }
  }
  if (have) load_b(0);
  if (linear && p.ksplit == 1 && PREFETCH) prefetch_epilogue(false);
  {
    // Row decode: lane j of a wave resolves the wave's row j & 31 ONCE (source block, top-left tap, image size); the 8 lanes that
    // gather the 8 k-chunks of a row then pick the record up with a lane shuffle (decoding per lane repeated the two divisions of a
    // row 8 times, four rows per lane)
    const int j = lane & 31;
    const int m = m0 + RPP * (j >> 3) + 8 * uw + (j & 7);
    const int sg = one_seg ? sg_first : (m < p.M ? seg_of(m) : 0);
    const Geo gl = one_seg ? gu : load_geo(sg);
    unsigned b, oy, ox; int cls;
    rowpos((unsigned)m - gl.mstart, gl, b, oy, ox, cls);
    const int d_hw = m < p.M ? (int)(gl.H | (gl.W << 16)) : 0;     // rows past M keep H = W = 0: every tap of theirs is out of the image
    const int d_base = (int)(gl.src0b + b * gl.imgb);
    const int d_y = p.transposed ? (int)oy + p.pad : (int)oy * p.stride - p.pad;
    const int d_x = p.transposed ? (int)ox + p.pad : (int)ox * p.stride - p.pad;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int src = i * 8 + prow;
      const int hw = __shfl(d_hw, src);
      rbase[i] = (unsigned)__shfl(d_base, src); ry0[i] = __shfl(d_y, src); rx0[i] = __shfl(d_x, src);
      rH[i] = hw & 0xffff; rW[i] = (int)((unsigned)hw >> 16);
    }
    if constexpr (X3) {
      if (tapin) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          tbase[i] = rbase[i] + (unsigned)(((ry0[i] * rW[i] + rx0[i]) * p.C + kc * 8) * 2);      // (modular: only used where the tap is valid)
          unsigned vm = 0;
          for (int tq = 0; tq < ntaps; ++tq) {
            const int r_ = tq / p.S, s_ = tq - r_ * p.S;
            const int y = p.transposed ? ry0[i] - r_ * p.dil : ry0[i] + r_ * p.dil, x = p.transposed ? rx0[i] - s_ * p.dil : rx0[i] + s_ * p.dil;
            vm |= ((unsigned)y < (unsigned)rH[i] && (unsigned)x < (unsigned)rW[i]) ? (1u << tq) : 0u;
          }
          tvm[i] = vm;
        }
      }
    }
  }
  TSTAMP(9);
  if (have) { load_a(0); advance(); }
  __syncthreads();
  if (!linear && p.ksplit == 1 && PREFETCH) prefetch_epilogue(true);     // destination rows are not consecutive: they come from the LDS table
  TSTAMP(2);
#ifdef AOD_TILE_TIMING
  if (g_tile_stamps && threadIdx.x == 0) g_tile_stamps[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memtime();     // shader clock around the K loop
#endif
  const int lr = lane & 15, lq = lane >> 4;
  // STAGGER (8-wave forms): waves 4-7 share their SIMDs with waves 0-3 and run the same program -- in lockstep both waves of a SIMD issue
  // their LDS-DMA, then both read fragments, then both want the matrix pipe.  Waves 4-7 therefore run half a K-step late: the MFMAs
  // of the second k-block are carried across the barrier (their fragments are already in registers) and issued at the top of the next
  // iteration, while waves 0-3 issue their loads and fragment reads (MI355X_MICROARCH.md 'Two waves per SIMD', item 9).
  const bool late = p.stagger && NT == 512 && uw >= 4;
  bool carried = false;
  bf16x8 af[MI], bfr[NI];
  auto frag_read = [&](const char* sa, const char* sb, int ks) {
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
  };
  auto mfma_block = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  };
  // one K-step.  Plain: the two 32-deep k-blocks.  X3: k-block 0 = heads, k-block 1 = tails of the same 32 channels -> xh*wh, xl*wh, xh*wl
  // (the tails of the activations are read while the head fragments stay in registers; the weight fragments are replaced last)
  // (`defer`: a late wave of the staggered 8-wave form leaves the LAST product of the step to the top of the next iteration, like the
  // plain form leaves its second k-block)
  auto kstep = [&](const char* sa, const char* sb, bool defer) {
    frag_read(sa, sb, 0);
    mfma_block();
    if constexpr (X3) {
      bf16x8 al[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 16 + lr;
        al[i] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + (((4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfr[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WN + j * 16 + lr;
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + (((4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
      if (!defer) mfma_block();
    } else {
      frag_read(sa, sb, 1);
      if (!defer) mfma_block();
    }
  };
  if (ST == 3) {
    // Three-stage ring (4-wave forms whose K-step is shorter than a load: one stage ahead, every K-step waits out the rest of a
    // load latency).  Stage s + 2 is issued at step s, the wait before the barrier names how many younger LDS-DMA instructions may
    // stay in flight (stage s + 2's: A_IT + B_IT per wave; loads, stores and LDS-DMA complete in issue order), and the barrier is a raw
    // s_barrier -- a __syncthreads() drains the queue.  The barrier also says that every wave is done reading stage s, whose buffer
    // stage s + 3 overwrites one step later.  Same K order as the two-stage loop: identical results.
    constexpr int LPS = A_IT + B_IT;
    bool have1 = have && __builtin_amdgcn_readfirstlane(kt_ld) < kt_end;
    if (have1) gload(1);
    int cur = 0;
    while (have) {
      const bool more2 = have1 && __builtin_amdgcn_readfirstlane(kt_ld) < kt_end;
      if (more2) gload(cur == 0 ? 2 : cur - 1);
      const char* sa = smem + cur * STAGE;
      const char* sb = sa + A_BYTES;
      kstep(sa, sb, false);
      if (!have1) break;
      __builtin_amdgcn_sched_barrier(0);
      if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == 2 ? 0 : cur + 1;
      have1 = more2;
    }
    __syncthreads();      // the epilogue image overlays the ring
  } else {
  for (int cur = 0; have; cur ^= 1) {
    const bool more = __builtin_amdgcn_readfirstlane(kt_ld) < kt_end;
    if (late && carried) mfma_block();
    if (more) gload(cur ^ 1);
    have = more;
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + A_BYTES;
    static_assert(BK == 64, "two 32-deep k-blocks per K-step");
    if constexpr (X3) { kstep(sa, sb, late); carried = late; }
    else {
      frag_read(sa, sb, 0);
      mfma_block();
      frag_read(sa, sb, 1);
      if (!late) mfma_block(); else carried = true;
    }
    __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier: next tile is resident afterwards
  }
  if (late && carried) mfma_block();
  }

  if (p.ksplit > 1) {
    // split-K: this slice's fp32 partial tile goes to its own slab, straight from the accumulators (one dword per lane, 16
    // consecutive columns per row group); no atomics, so the finalize pass can add the slabs in a fixed order
    float* const slab = p.ws + (long long)kz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * WM + i * 16 + lq * 4 + r, n = n0 + wn * WN + j * 16 + lr;
          if (m < p.M && n < p.N) slab[(long long)m * p.N + n] = acc[i][j][r];
        }
    return;
  }
  // ---- epilogue: accumulators -> LDS (fp32, [BM][CP]) -> row-major vector stores
#ifdef AOD_TILE_TIMING
  if (g_tile_stamps && threadIdx.x == 0) g_tile_stamps[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memtime();
#endif
  TSTAMP(3);
  if (!PREFETCH) prefetch_epilogue(!linear);     // big tile: the bias / scale vectors are fetched after the K loop (16-24 registers it cannot spare)
  float* sc = reinterpret_cast<float*>(smem);
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
  const int r0e = ep * EBM;                  // first tile row of this pass
  if (EP > 1 && ep > 0) __syncthreads();     // the previous pass's image has been stored
  if (EP == 1 || (wm * WM) / EBM == ep) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sc[(wm * WM - r0e + i * 16 + lq * 4 + r) * CP + wn * WN + j * 16 + lr] = acc[i][j][r];
  }
  if (LATE && fast && OPS > 0 && ecok) {
#pragma unroll
    for (int it = 0; it < E_IT; ++it) {
      const long long eoff = linear ? lin_off + (long long)r0e * NP + it * lin_step : s_drow[r0e + er + it * (NT / NCHT)] * NP + xcol(n0 + ec * 8);
      if (OPS > 1 && p.res) {
        lres[it] = *reinterpret_cast<const bf16x8*>(p.res + eoff);
        if constexpr (X3) lrl[it] = *reinterpret_cast<const bf16x8*>(p.res + eoff + 32);
      }
      if (g_mask) lmask[it] = *reinterpret_cast<const bf16x8*>(g_mask + eoff);
    }
  }
  __syncthreads();
  TSTAMP(4);

  // fast tiles: no per-thread predicates at all, only workgroup-uniform branches -- the general loop below costs ~500 instructions
  // per 16-B store, this one under 100
  if (fast) {
    if (ecok) {
    bf16_t* const yb = reinterpret_cast<bf16_t*>(g_y) + xcol(n0 + ec * 8);
    bf16_t* const yl = reinterpret_cast<bf16_t*>(g_y) + lin_off + (long long)r0e * NP;
#pragma unroll
    for (int it = 0; it < E_IT; ++it) {
      const int row = er + it * (NT / NCHT);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8 + 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
      if (p.pre_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= cs1[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += cb1[j];
      const long long eoff = linear ? lin_off + (long long)r0e * NP + it * lin_step : s_drow[r0e + row] * NP + xcol(n0 + ec * 8);
      if (OPS > 1 && p.res) {
        const bf16x8 rv = PREFETCH ? pres[it] : (LATE ? lres[it] : *reinterpret_cast<const bf16x8*>(p.res + eoff));
        if constexpr (X3) {
          const bf16x8 rl = LATE ? lrl[it] : *reinterpret_cast<const bf16x8*>(p.res + eoff + 32);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j] + (float)rl[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
        }
      }
      if (OPS > 0 && g_mask) {
        const bf16x8 mv = PREFETCH ? pmask[it] : (LATE ? lmask[it] : *reinterpret_cast<const bf16x8*>(g_mask + eoff));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)mv[j] > 0.f) ? v[j] : 0.f;
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) { csum[j] += v[j]; ov[j] = (bf16_t)v[j]; }
      bf16_t* const yo = linear ? yl + it * lin_step : yb + s_drow[r0e + row] * NP;
      *reinterpret_cast<bf16x8*>(yo) = ov;
      if constexpr (X3) {
        bf16x8 ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) ol[j] = (bf16_t)(v[j] - (float)ov[j]);
        *reinterpret_cast<bf16x8*>(yo + 32) = ol;
      }
    }
    }
  } else if (p.out_f32 && !p.res && !g_mask && !p.zraw && !p.post_scale && !p.pre_scale && (p.N & 3) == 0) {
    // fp32 destinations of the prediction convs (bias, optional ReLU, no other operand; N = 180 / 36 / 720 ...): rows and 4-column halves are
    // the only predicates -- the general loop below spends ~500 instructions per 16-B store on operands these launches do not have.
    // (v * 1 + bias of the general loop = v + bias: the same bits)
    const int n = n0 + ec * 8;
    if (ecok && n < p.N) {
      const bool hi = n + 8 <= p.N;                 // else columns n .. n + 3 only (N % 4 == 0)
      float* const yo = reinterpret_cast<float*>(g_y) + n;
#pragma unroll
      for (int it = 0; it < E_IT; ++it) {
        const int row = er + it * (NT / NCHT);
        if (m0 + r0e + row >= p.M) continue;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = v0[j] * 1.0f + cb1[j]; v1[j] = v1[j] * 1.0f + cb1[4 + j]; }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { csum[j] += v0[j]; if (hi) csum[4 + j] += v1[j]; }
        float* const o = yo + s_drow[r0e + row] * p.N;
        *reinterpret_cast<f32x4*>(o) = v0;
        if (hi) *reinterpret_cast<f32x4*>(o + 4) = v1;
      }
    }
  } else
#pragma unroll
  for (int it = 0; it < E_IT; ++it) {
    const int row = er + it * (NT / NCHT);
    const int m = m0 + r0e + row;
    const int n = n0 + ec * 8;
    if (m >= p.M || n >= p.N || !ecok) continue;
    const long long drow = s_drow[r0e + row];
    float v[8], raw[8];
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(sc + row * CP + ec * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    const bool full = (n + 8 <= p.N);
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = v[j];
    if (full) {
      const long long off = drow * NP + xcol(n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * cs1[j] + cb1[j];
      if (OPS > 1 && p.res) {
        const bf16x8 rv = PREFETCH ? pres[it] : *reinterpret_cast<const bf16x8*>(p.res + off);
        if (xout) {
          // v + (head + tail), the association of the fast path above (and of the fused bottleneck kernels): a ragged tile must not round an
          // element differently from an interior one -- (v + head) + tail differs in the last bit for ~0.2 % of the elements, which made a
          // row's bits depend on where the tile boundaries fall (i.e. on the batch size)
          const bf16x8 rl = *reinterpret_cast<const bf16x8*>(p.res + off + 32);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j] + (float)rl[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
        }
      }
      if (OPS > 0 && g_mask) {
        const bf16x8 mv = PREFETCH ? pmask[it] : *reinterpret_cast<const bf16x8*>(g_mask + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)mv[j] > 0.f) ? v[j] : 0.f;
      }
      if (p.post_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= cs2[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) csum[j] += v[j];
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(g_y) + off;
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
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(g_y) + off) = ov;
        if (xout) {
          bf16x8 ol;
#pragma unroll
          for (int j = 0; j < 8; ++j) ol[j] = (bf16_t)(v[j] - (float)ov[j]);
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(g_y) + off + 32) = ol;
        }
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
        if (g_shift) u += g_shift[n + j];
        if (p.res) u += (float)p.res[off];
        if (g_mask) u = ((float)g_mask[off] > 0.f) ? u : 0.f;
        if (p.post_scale) u *= p.post_scale[n + j];
        if (p.relu) u = fmaxf(u, 0.f);
        csum[j] += u;
        if (p.out_f32) reinterpret_cast<float*>(g_y)[off] = u;
        else reinterpret_cast<bf16_t*>(g_y)[off] = (bf16_t)u;
        if (p.zraw) p.zraw[off] = (bf16_t)raw[j];
      }
    }
  }
  }    // epilogue passes
  TSTAMP(5);
#ifdef AOD_TILE_TIMING
  __builtin_amdgcn_s_waitcnt(0);      // vmcnt/lgkmcnt 0: stores acknowledged
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSTAMP(6);
  if (g_tile_stamps && threadIdx.x == 0)
    g_tile_stamps[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#endif
  if (g_colsum) {
    // column sums of this tile: per-thread partials -> LDS [NT / NCHT][BN] -> one fp32 atomic per column
    __syncthreads();
    float* sr = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (ecok) sr[er * BN + ec * 8 + j] = csum[j];
    __syncthreads();
    if (t < BN && n0 + t < p.N) {
      float s = 0.f;
      for (int r = 0; r < NT / NCHT; ++r) s += sr[r * BN + t];
      if (p.cs_ws) p.cs_ws[((long long)gi * p.tiles_m + tile_m) * p.N + n0 + t] = s;
      else atomicAdd(g_colsum + n0 + t, s);
    }
  }
}

#undef g_y
#undef g_mask
#undef g_colsum
#undef AOD_GSEL

template <int BM, int BN, int NT = 256, int OPS = 2, bool GROUPED = false, int ST = 2, bool X3 = false>
static int launch_conv(const ConvKParams& p, hipStream_t st) {
  ConvKParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  const size_t stage = (size_t)(BM + BN) * 128 * ST;
  const size_t epi_full = (size_t)BM * (BN + 4) * 4;
  const size_t epi = epi_full > 144 * 1024 ? epi_full / 2 : epi_full;      // (two epilogue passes for the 256 x 256 tile)
  const size_t lds = (stage > epi ? stage : epi) + (size_t)BM * 8;     // staging | fp32 epilogue image, then the destination-row table
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, NT, OPS, GROUPED, ST, X3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  // deterministic mode: partial column sums per (group, row tile) into the scratch, added in order below (a split-K launch has no epilogue
  // of its own: conv_splitk_finalize_kernel takes the scratch)
  bool any_cs = q.colsum != nullptr;
  if (GROUPED) for (int g = 0; g < q.ngroups; ++g) any_cs = any_cs || q.grp[g].colsum;
  const int ng = GROUPED ? q.ngroups : 1;
  if (q.ksplit == 1) q.cs_ws = any_cs ? aod_det_scratch((size_t)ng * q.tiles_m * q.N) : nullptr;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, NT, OPS, GROUPED, ST, X3>), dim3(q.tiles_m * q.tiles_n * (q.ngroups > 1 ? q.ngroups : q.ksplit)), dim3(NT), lds, st, q);
  if (q.ksplit == 1 && q.cs_ws) {
    for (int g = 0; g < ng; ++g) {
      float* dst = GROUPED ? q.grp[g].colsum : q.colsum;
      if (dst) { const int rc = aod_colsum_finalize(q.cs_ws + (size_t)g * q.tiles_m * q.N, q.tiles_m, q.N, q.N, dst, nullptr, 0, st); if (rc) return rc; }
    }
  }
  return 0;
}

static int fill_params(const aod_conv_desc_t* d, ConvKParams& p) {
  AOD_CHECK_ARG(d->nseg >= 1 && d->nseg <= 8, "conv: nseg %d out of range", d->nseg);
  AOD_CHECK_ARG(d->C % 8 == 0, "conv: source channels %d must be a multiple of 8", d->C);
  AOD_CHECK_ARG(d->stride >= 1 && d->dil >= 1 && d->R >= 1 && d->S >= 1, "conv: bad geometry");
  AOD_CHECK_ARG(d->x3 == 0 || d->x3 == 1, "conv: aod_conv_desc_t.x3 = %d (0 or 1; descriptors must be zero-initialised -- the field took the place of padding in aod_version() 2)", d->x3);
  p.C = d->C; p.N = d->N; p.R = d->R; p.S = d->S; p.K = d->R * d->S * d->C;
  p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.transposed = d->transposed;
  p.relu = d->relu; p.out_f32 = d->out_f32; p.nseg = d->nseg; p.x3 = d->x3 ? 1 : 0;
  if (p.x3) AOD_CHECK_ARG(d->C % 64 == 0, "conv (x3): X-layout source width %d must be a multiple of 64", d->C);
  long long m = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const aod_conv_seg_t& s = d->seg[i];
    if (!d->transposed) {
      const int eh = (s.H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
      const int ew = (s.W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
      // (a destination smaller than the natural output is its top-left part: the space-to-depth stem computes 256 of the 257 rows /
      // columns a 4x4 / pad-2 filter would give)
      AOD_CHECK_ARG(s.OH >= 1 && s.OW >= 1 && s.OH <= eh && s.OW <= ew, "conv: segment %d output %dx%d exceeds the expected %dx%d", i, s.OH, s.OW, eh, ew);
    } else {
      const int eh = (s.OH + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
      const int ew = (s.OW + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
      AOD_CHECK_ARG(eh == s.H && ew == s.W, "dgrad: segment %d dZ %dx%d != expected %dx%d", i, s.H, s.W, eh, ew);
    }
    AOD_CHECK_ARG(s.H < 65536 && s.W < 65536 && s.OH < 65536 && s.OW < 65536, "conv: segment %d image side >= 65536", i);
    p.segB[i] = s.B; p.segH[i] = s.H; p.segW[i] = s.W; p.segOH[i] = s.OH; p.segOW[i] = s.OW;
    p.seg_src0[i] = s.src_row0; p.seg_dst0[i] = s.dst_row0;
    m += (long long)s.B * s.OH * s.OW;
    AOD_CHECK_ARG(m < (1ll << 31), "conv: too many rows");
    p.seg_mend[i] = (int)m;
  }
  for (int i = d->nseg; i < 8; ++i) { p.segB[i] = p.segH[i] = p.segW[i] = p.segOH[i] = p.segOW[i] = 0; p.seg_src0[i] = p.seg_dst0[i] = 0; p.seg_mend[i] = (int)m; }
  p.M = (int)m;
  p.bigrows = 0;
  for (int i = 0; i < d->nseg; ++i)
    if ((long long)d->seg[i].B * d->seg[i].OH * d->seg[i].OW >= (1ll << 22)) p.bigrows = 1;
  return 0;
}

// Split-K second pass: the epilogue of conv_igemm_kernel applied to the fp32 sums in the workspace.  A block takes 8 GEMM rows x 256
// columns (thread -> one 8-column chunk of one row; the outputs are tiny, the grid has to be wide), column sums need one LDS
// reduction and 256 atomics per block.
__global__ __launch_bounds__(256) void conv_splitk_finalize_kernel(const ConvKParams p) {
  __shared__ float sr[8][256];
  const int t = threadIdx.x, ec = t & 31, er = t >> 5;
  const int n = (blockIdx.y * 32 + ec) * 8;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float cs1[8], cb1[8], cs2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = n + j < p.N;
    cs1[j] = (p.pre_scale && ok) ? p.pre_scale[n + j] : 1.f;
    cb1[j] = (p.pre_shift && ok) ? p.pre_shift[n + j] : 0.f;
    cs2[j] = (p.post_scale && ok) ? p.post_scale[n + j] : 1.f;
  }
  const bool vec = (p.N & 7) == 0 && n + 8 <= p.N;
  if (n < p.N) {
    for (int k = 0; k < 1; ++k) {
      const int m = blockIdx.x * 8 + er;
      if (m >= p.M) break;
      int sg = 0;
      if (p.nseg > 1) {
#pragma unroll
        for (int q = 0; q < 7; ++q) sg += (m >= p.seg_mend[q]) ? 1 : 0;
      }
      const long long drow = p.seg_dst0[sg] + (m - (sg ? p.seg_mend[sg - 1] : 0));
      const float* w = p.ws + (long long)m * p.N + n;
      float raw[8], v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = 0.f;
      const long long slab = (long long)p.M * p.N;
      for (int z = 0; z < p.ksplit; ++z) {           // fixed order: the sum does not depend on which slice finished first
        if (vec) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(w + z * slab), b = *reinterpret_cast<const f32x4*>(w + z * slab + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { raw[j] += a[j]; raw[4 + j] += b[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) raw[j] += n + j < p.N ? w[z * slab + j] : 0.f;
        }
      }
      // (X3 launches with a bf16 destination: X-layout rows, heads and tails 32 columns apart)
      const bool xout = p.x3 && !p.out_f32;
      const int NP = xout ? ((p.N + 31) >> 5) << 6 : p.N;
      const long long off = drow * NP + (xout ? ((n >> 5) << 6) + (n & 31) : n);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (n + j >= p.N) { v[j] = 0.f; continue; }
        float u = raw[j] * cs1[j] + cb1[j];
        if (p.res) u += xout ? (float)p.res[off + j] + (float)p.res[off + 32 + j] : (float)p.res[off + j];
        if (p.mask) u = ((float)p.mask[off + j] > 0.f) ? u : 0.f;
        if (p.post_scale) u *= cs2[j];
        if (p.relu) u = fmaxf(u, 0.f);
        v[j] = u;
        csum[j] += u;
      }
      if (vec && !p.out_f32) {
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + off) = ov;
        if (xout) {
          bf16x8 ol;
#pragma unroll
          for (int j = 0; j < 8; ++j) ol[j] = (bf16_t)(v[j] - (float)ov[j]);
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + off + 32) = ol;
        }
      } else {
        for (int j = 0; j < 8 && n + j < p.N; ++j) {
          if (p.out_f32) reinterpret_cast<float*>(p.y)[off + j] = v[j];
          else reinterpret_cast<bf16_t*>(p.y)[off + j] = (bf16_t)v[j];
        }
      }
      if (p.zraw)
        for (int j = 0; j < 8 && n + j < p.N; ++j) p.zraw[off + j] = (bf16_t)raw[j];
    }
  }
  if (p.colsum) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sr[er][ec * 8 + j] = csum[j];
    __syncthreads();
    const int c = blockIdx.y * 256 + t;
    if (c < p.N) {
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s2 += sr[r][t];
      if (p.cs_ws) p.cs_ws[(long long)blockIdx.x * p.N + c] = s2;
      else atomicAdd(p.colsum + c, s2);
    }
  }
}

// Split-K pays when the output is so small that whole tiles cannot fill the device: few tiles x very many K-steps (pyramid level P6:
// a stride-2 3x3 on the 2048-channel C5, 8 x 8 outputs per image; the 3x3 convs of the 16 x 16 backbone stage, K = 4608).  The decision and the slice boundaries depend on the PER-IMAGE
// geometry and on K only, never on the batch size: the fp32 summation order of an output element -- and so the score of an image --
// must not change with how the pool is batched.  Returns the number of K slices (1 = direct launch).
// operands / geometry of a launch in the persistent kernel's terms (conv_x3p.hip)
static void x3p_args(const ConvKParams& p, X3PArgs& a) {
  memset(&a, 0, sizeof(a));
  a.x = p.x; a.w = p.w; a.y = reinterpret_cast<bf16_t*>(p.y); a.pre_scale = p.pre_scale; a.pre_shift = p.pre_shift; a.res = p.res; a.mask = p.mask;
  a.colsum = p.colsum; a.C = p.C; a.N = p.N; a.K = p.K; a.taps = p.R * p.S; a.S = p.S; a.stride = p.stride; a.pad = p.pad; a.dil = p.dil;
  a.transposed = p.transposed; a.relu = p.relu; a.nseg = p.nseg; a.M = p.M; a.x_bytes = p.x_bytes; a.w_bytes = p.w_bytes;
  a.tapin = (p.tap_inner && a.taps > 1 && p.C >= 256) ? 1 : 0;      // the general kernel's K order for this shape (see `tapin` there)
  for (int i = 0; i < 8; ++i) {
    a.segH[i] = p.segH[i]; a.segW[i] = p.segW[i]; a.segOH[i] = p.segOH[i]; a.segOW[i] = p.segOW[i]; a.segB[i] = p.segB[i];
    a.seg_src0[i] = p.seg_src0[i]; a.seg_dst0[i] = p.seg_dst0[i]; a.seg_mend[i] = p.seg_mend[i];
  }
  if (p.ngroups > 1) {
    a.ngroups = p.ngroups;
    for (int g = 0; g < p.ngroups; ++g) {
      a.grp[g].x = p.grp[g].x; a.grp[g].w = p.grp[g].w; a.grp[g].y = reinterpret_cast<bf16_t*>(p.grp[g].y); a.grp[g].shift = p.grp[g].pre_shift;
      a.grp[g].mask = p.grp[g].mask; a.grp[g].colsum = p.grp[g].colsum;
    }
  }
}

static int choose_ksplit(const ConvKParams& p) {
  const int nk = (p.K + 63) / 64;
  if (nk < 64) return 1;
  static const char* dbg_hw = getenv("AOD_KSPLIT_MAXHW");      // (debug: the largest per-image output that is still split)
  const long long maxhw = dbg_hw ? atoll(dbg_hw) : 256;
  long long rows16 = 0;                 // GEMM rows of a NOMINAL batch of 16 images (the batch size itself must not enter: see above)
  for (int i = 0; i < p.nseg; ++i) {
    if (p.segB[i] <= 0) continue;
    if ((long long)p.segOH[i] * p.segOW[i] > maxhw) return 1;
    rows16 += 16ll * p.segOH[i] * p.segOW[i];
  }
  static const char* dbg_ks = getenv("AOD_KSPLIT_STEPS");      // (debug: a fixed number of K-steps per slice, the rule of the earlier rounds at 16)
  if (dbg_ks) { const int per = atoi(dbg_ks); return nk / per > 1 ? nk / per : 1; }
  // as many slices as fill ONE round of two workgroups per CU with the split launch's 128-row tiles -- one slice more starts a second,
  // nearly empty round (P6 at 16 x 8 x 8: 36 slices 66.7 us, 24 slices 54.4 us) -- but no slice shorter than 8 K-steps
  const long long tiles = ((rows16 + 127) / 128) * ((p.N + 127) / 128);
  long long ks = 512 / (tiles > 0 ? tiles : 1);
  if (ks > nk / 8) ks = nk / 8;
  return ks > 1 ? (int)ks : 1;
}

static int conv_params(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst, const float* pre_scale,
                       const float* pre_shift, const void* res, const void* mask, const float* post_scale, void* zraw, float* colsum,
                       ConvKParams& p) {
  AOD_CHECK_ARG(desc && src && w_packed && dst, "conv: null pointer");
  memset(&p, 0, sizeof(p));
  int rc = fill_params(desc, p);
  if (rc) return rc;
  AOD_CHECK_ARG(!(desc->out_f32 && zraw), "conv: zraw needs a bf16 destination");
  if (desc->x3) {
    AOD_CHECK_ARG(!zraw && !post_scale, "conv (x3): zraw / post_scale are not supported");
    AOD_CHECK_ARG(desc->out_f32 || desc->N % 32 == 0, "conv (x3): a bf16 (X-layout) destination needs N %% 32 == 0, got %d", desc->N);
    AOD_CHECK_ARG(!(desc->out_f32 && (res || mask)), "conv (x3): residual / mask operands need an X-layout destination");
  }
  p.x = (const bf16_t*)src; p.w = (const bf16_t*)w_packed; p.y = dst;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.res = (const bf16_t*)res; p.mask = (const bf16_t*)mask;
  p.post_scale = post_scale; p.zraw = (bf16_t*)zraw; p.colsum = colsum;
  p.ksplit = 1;
  { static const char* dbg_st = getenv("AOD_STAGGER"); p.stagger = (dbg_st && dbg_st[0] == '0') ? 0 : 1; }
  { static const char* dbg_ti = getenv("AOD_X3_TAPS_INNER"); p.tap_inner = (dbg_ti && dbg_ti[0] == '0') ? 0 : 1; }     // (debug: 0 = the plain K order)
  p.perm = (desc->transposed && desc->stride == 2 && desc->R * desc->S > 1 && desc->R * desc->S <= 64) ? 1 : 0;     // (no gain measured for 1x1)
  // The dgrad of a 1x1 / stride-2 conv ACCUMULATED IN PLACE (res == dst, nothing else in the epilogue: the running sum of a gradient
  // junction, functional.GradAcc) only changes the (even, even) pixels of the destination: dX[b, 2y, 2x] += dZ[b, y, x] . W.  As a general
  // transposed launch it walks K for all four pixel classes and rewrites 4 x the rows (206 us for the layer-3 entry at 16 x 64 x 64 x 512
  // in the reference-precision mode); as a LATTICE launch it is a plain GEMM over the dZ pixels whose rows are stored two apart.
  if (desc->transposed && desc->stride == 2 && desc->R == 1 && desc->S == 1 && desc->pad == 0 && desc->nseg == 1 && res && res == dst &&
      !mask && !colsum && !pre_scale && !pre_shift && !post_scale && !zraw && !desc->relu && !desc->out_f32 && p.M > 0) {
    static const char* dbg_lat = getenv("AOD_DGRAD_LATTICE");
    if (!(dbg_lat && dbg_lat[0] == '0')) {
      p.up_w = p.segOW[0]; p.up_hw = p.segOH[0] * p.segOW[0];
      p.segOH[0] = p.segH[0]; p.segOW[0] = p.segW[0];
      p.stride = 1;
      const long long m = (long long)p.segB[0] * p.segH[0] * p.segW[0];
      p.M = (int)m;
      for (int i = 0; i < 8; ++i) p.seg_mend[i] = (int)m;
      p.bigrows = m >= (1ll << 22) ? 1 : 0;
    }
  }
  static const char* dbg_perm = getenv("AOD_DGRAD_CLASSES");
  if (dbg_perm && dbg_perm[0] == '0') p.perm = 0;
  long long xrows = 0;
  for (int i = 0; i < desc->nseg; ++i) {
    const aod_conv_seg_t& sg = desc->seg[i];
    AOD_CHECK_ARG(sg.src_row0 >= 0, "conv: negative src_row0");
    const long long e = sg.src_row0 + (long long)sg.B * sg.H * sg.W;
    if (e > xrows) xrows = e;
  }
  p.x_bytes = xrows * p.C * 2;
  p.w_bytes = (long long)p.N * p.K * 2;
  AOD_CHECK_ARG(p.x_bytes < 0xe0000000ll && p.w_bytes < 0xe0000000ll, "conv: operand larger than 3.5 GiB (32-bit buffer offsets)");
  return 0;
}

extern "C" size_t aod_conv2d_ws_bytes(const aod_conv_desc_t* desc) {
  ConvKParams p;
  memset(&p, 0, sizeof(p));
  if (!desc || fill_params(desc, p) != 0 || p.M == 0) return 0;
  const int ks = choose_ksplit(p);
  return ks > 1 ? (size_t)ks * p.M * p.N * 4 : 0;
}

extern "C" int aod_conv2d_ws(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst,
                             const float* pre_scale, const float* pre_shift, const void* res, const void* mask,
                             const float* post_scale, void* zraw, float* colsum, void* workspace, size_t workspace_bytes,
                             aod_stream_t stream) {
  ConvKParams p;
  int rc = conv_params(desc, src, w_packed, dst, pre_scale, pre_shift, res, mask, post_scale, zraw, colsum, p);
  if (rc) return rc;
  if (p.M == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // 1x1, stride 1: a plain GEMM over consecutive rows -- the persistent streaming kernel (pointwise.hip) when its launch heuristic wants it
  if (!p.x3 && !p.up_w && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.nseg == 1 && !p.out_f32 && !p.zraw && !p.post_scale) {
    PwArgs a;
    const long long s0 = p.seg_src0[0], d0 = p.seg_dst0[0];
    a.x = p.x + s0 * p.C; a.w = p.w; a.y = reinterpret_cast<bf16_t*>(p.y) + d0 * p.N;
    a.pre_scale = p.pre_scale; a.pre_shift = p.pre_shift;
    a.res = p.res ? p.res + d0 * p.N : nullptr; a.mask = p.mask ? p.mask + d0 * p.N : nullptr; a.colsum = p.colsum;
    a.M = p.M; a.N = p.N; a.K = p.C; a.relu = p.relu;
    if (aod_pw_wants(a)) return aod_pw_gemm(a, st);
  }
  const int ks = (workspace && !p.perm && !p.up_w) ? choose_ksplit(p) : 1;
  if (ks > 1) {
    const size_t need = (size_t)ks * p.M * p.N * 4;
    AOD_CHECK_ARG(workspace_bytes >= need, "conv: split-K workspace of %zu bytes, need %zu (aod_conv2d_ws_bytes)", workspace_bytes, need);
    p.ksplit = ks; p.ws = (float*)workspace;
    if (p.x3) { if (p.N > 64) launch_conv<128, 128, 256, 2, false, 2, true>(p, st); else launch_conv<128, 64, 256, 2, false, 2, true>(p, st); }
    else if (p.N > 64) launch_conv<128, 128>(p, st); else launch_conv<128, 64>(p, st);
    AOD_LAUNCH_CHECK();
    p.cs_ws = p.colsum ? aod_det_scratch((size_t)((p.M + 7) / 8) * p.N) : nullptr;
    hipLaunchKernelGGL(conv_splitk_finalize_kernel, dim3((p.M + 7) / 8, (p.N + 255) / 256), dim3(256), 0, st, p);
    AOD_LAUNCH_CHECK();
    if (p.cs_ws) return aod_colsum_finalize(p.cs_ws, (p.M + 7) / 8, p.N, p.N, p.colsum, nullptr, 0, st);
    return 0;
  }
  // tile choice: the largest tile that still gives >= 2 workgroups per CU (2 x 256); else the most workgroups
  auto ntiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  static const char* dbg_want = getenv("AOD_TILE_WANT");           // (debug: the tile count that counts as 'fills the device')
  const long long want = dbg_want ? atoll(dbg_want) : 512;
  // a ragged last column tile (N = 180 -> 128 + 52) wastes MFMA work; 64-wide tiles trim it (192 instead of 256 columns)
  const int pad128 = (p.N + 127) / 128 * 128, pad64 = (p.N + 63) / 64 * 64;
  const bool ragged = p.N > 128 && (pad128 - pad64) * 5 >= pad128;
  if (p.x3 && !p.out_f32 && !p.zraw && !p.post_scale && !p.bigrows && p.R == p.S) {
    // the persistent producer / consumer kernel (conv_x3p.hip) for the 128-column-tileable 1x1 / 3x3 layers that fill at least ~ a round of
    // the CUs with 128 x 128 tiles and do NOT qualify for the 256 x 256 tile below (the head towers, FPN P3): identical bits, AOD_X3P=0 disables
    const long long t256 = ntiles(256, 256);
    const bool big = !p.res && p.N % 256 == 0 && p.K >= 2048 && t256 >= 240 && t256 * 100 >= ((t256 + 255) / 256) * 256 * 92;
    const char* dbg_big = getenv("AOD_X3P_OVER_256");                  // (A/B, read per call: also take the launches of the 256 x 256 tile)
    if (!big || (dbg_big && dbg_big[0] == '1')) {
      X3PArgs a;
      x3p_args(p, a);
      // (the class-major stride-2 dgrad and the in-place 1x1 / stride-2 dgrad as lattice launches of the persistent kernel: X3PArgs.lat)
      a.lat = p.perm ? 1 : (p.up_w ? 2 : 0);
      a.up_w = p.up_w; a.up_hw = p.up_hw;
      if (aod_conv_x3p_wants(a, p.colsum && aod_get_deterministic())) return aod_conv_x3p_launch(a, st);
    }
  }
  if (p.x3) {
    // X3: the 256 x 256 tile under the same fill rule as the plain form (a K-step of the 128 x 128 tile asks the L2 -> LDS path for 64 KB
    // per 1 536 matrix-pipe cycles per CU -- more than it delivers --, the big tile for half of that), else the 4-wave forms
    {
      static const char* dbg_t256x = getenv("AOD_TILE_256");
      const long long t256 = ntiles(256, 256);
      const bool fits = t256 >= 240 && t256 * 100 >= ((t256 + 255) / 256) * 256 * 92;
      if (!(dbg_t256x && dbg_t256x[0] == '0') && !p.res && !p.out_f32 && p.N % 256 == 0 && p.K >= 2048 && fits) {
        if (!p.mask) launch_conv<256, 256, 512, 0, false, 2, true>(p, st); else launch_conv<256, 256, 512, 1, false, 2, true>(p, st);
        AOD_LAUNCH_CHECK();
        return 0;
      }
    }
    // 129 .. 192 output columns without epilogue operands (retina_cls, 180 columns, fp32 destination): a 192 x 192 tile on eight waves, one
    // workgroup per CU like the 256 x 256 tile (96 FLOP per staged byte; 64-wide tiles: 43, and the L2 -> LDS path bounds the 4-wave x3
    // forms), same K order -> same bits.  AOD_X3_TILE_192=0: the 128 x 64 tile.
    {
      const char* dbg_t192 = getenv("AOD_X3_TILE_192");      // (read per call: tests switch it in-process)
      const long long t192 = ntiles(192, 192);
      if (!(dbg_t192 && dbg_t192[0] == '0') && p.N > 128 && p.N <= 192 && !p.res && !p.mask && !p.zraw && p.K >= 2048 && t192 >= 240) {
        launch_conv<192, 192, 512, 0, false, 2, true>(p, st);
        AOD_LAUNCH_CHECK();
        return 0;
      }
    }
    // A/B knob (AOD_X3_128_W8=1): the 128 x 128 x3 tile on EIGHT waves for launches without a residual operand.  Back to back on warm operands
    // it beats the 64 x 128 three-stage tile on the 16 384-row layers (80.0 -> 73.2 us for the stage-3 3 x 3, tools/dbg/x3_w8_ab.sh); INSIDE the
    // step, on operands that come from HBM / the Infinity Cache, it loses 8 % (80.2 -> 86.3 us forward, 77.9 -> 89.3 us dgrad in the instrumented
    // step, profiles/r05_conv_shapes_one_step.txt against the run before it): not taken.
    static const char* dbg_w8x = getenv("AOD_X3_128_W8");
    const long long t128 = ntiles(128, 128);
    const bool w8 = dbg_w8x && dbg_w8x[0] == '1' && !p.res && !p.out_f32 && p.N % 128 == 0 && t128 >= 256;
    if (w8) {
      if (!p.mask) launch_conv<128, 128, 512, 0, false, 2, true>(p, st); else launch_conv<128, 128, 512, 1, false, 2, true>(p, st);
      AOD_LAUNCH_CHECK();
      return 0;
    }
    static const char* dbg_st3 = getenv("AOD_X3_128_ST3");        // (debug / A-B: the 128 x 128 x3 tile on a three-stage ring, one workgroup per CU)
    if (ragged && ntiles(128, 64) >= want) launch_conv<128, 64, 256, 2, false, 2, true>(p, st);
    else if (p.N > 64 && dbg_st3 && dbg_st3[0] == '1' && ntiles(128, 128) >= 256) launch_conv<128, 128, 256, 2, false, 3, true>(p, st);
    else if (p.N > 64 && ntiles(128, 128) >= want) launch_conv<128, 128, 256, 2, false, 2, true>(p, st);
    else if (p.N > 64 && ntiles(64, 128) >= want) launch_conv<64, 128, 256, 2, false, 3, true>(p, st);
    else if (p.N <= 64 && ntiles(128, 64) >= want) launch_conv<128, 64, 256, 2, false, 2, true>(p, st);
    else if (p.N > 64 && ntiles(128, 64) >= want && p.N % 128 != 0) launch_conv<128, 64, 256, 2, false, 2, true>(p, st);
    else launch_conv<64, 64, 256, 2, false, 3, true>(p, st);
    AOD_LAUNCH_CHECK();
    return 0;
  }
  // deep convs without a residual operand (forward and dgrad of the head towers, the 3x3 of the backbone): the 128 x 128 tile on 8
  // waves -- four waves per SIMD hide more of the K loop's waits than two (-4 % on the head-tower shape); with the residual's prefetch
  // registers as well the 8-wave form spills and loses
  // The 256 x 256 tile (one 8-wave workgroup per CU, 128 FLOP per staged byte instead of 64, two epilogue passes) runs deep-K layers
  // 15-18 % faster per tile (tools/dbg/tile256.py: 794 -> 934 TFLOP/s on the FPN P3 shape) but one workgroup per CU quantises hard: it
  // is chosen only when its tiles fill whole rounds of the 256 CUs to >= 92 % (AOD_TILE_256=1 forces it, =0 disables it).
  static const char* dbg_t256 = getenv("AOD_TILE_256");
  const long long t256 = ntiles(256, 256);
  const bool fits256 = t256 >= 240 && t256 * 100 >= ((t256 + 255) / 256) * 256 * 92;
  if (!(dbg_t256 && dbg_t256[0] == '0') && !p.res && !p.out_f32 && p.N % 256 == 0 && p.K >= 1024 &&
      ((dbg_t256 && dbg_t256[0] == '1' && t256 >= 128) || fits256)) {
    if (!p.mask) launch_conv<256, 256, 512, 0>(p, st); else launch_conv<256, 256, 512, 1>(p, st);
    AOD_LAUNCH_CHECK();
    return 0;
  }
  static const char* dbg_w8 = getenv("AOD_TILE_W8");
  if (!(dbg_w8 && dbg_w8[0] == '0') && !p.res && p.N >= 128 && p.K >= 1024 && ntiles(128, 128) >= want) {
    if (!p.mask) launch_conv<128, 128, 512, 0>(p, st); else launch_conv<128, 128, 512, 1>(p, st);
    AOD_LAUNCH_CHECK();
    return 0;
  }
  // (three LDS stages for the 64-row 4-wave tiles when the K loop is long enough to matter: AOD_RING3=0 disables; the 128 x 64 tile keeps
  // two -- three stages cost it its third workgroup per CU: 61 -> 69 us on the narrow prediction convs)
  static const char* dbg_r3 = getenv("AOD_RING3");
  const bool r3 = !(dbg_r3 && dbg_r3[0] == '0') && p.K >= 256;
  if (ragged && ntiles(128, 64) >= want) launch_conv<128, 64>(p, st);
  else if (p.N > 64 && ntiles(128, 128) >= want) launch_conv<128, 128>(p, st);      // (three stages = 96 KB = one workgroup per CU: measured +0.55 ms per step; 64 x 128 x 3 stages instead: +0.3 ms)
  else if (p.N > 64 && ntiles(64, 128) >= want) { if (r3) launch_conv<64, 128, 256, 2, false, 3>(p, st); else launch_conv<64, 128>(p, st); }
  else if (p.N <= 64 && ntiles(128, 64) >= want) launch_conv<128, 64>(p, st);
  else if (p.N > 64 && ntiles(128, 64) >= want && p.N % 128 != 0) launch_conv<128, 64>(p, st);
  else { if (r3) launch_conv<64, 64, 256, 2, false, 3>(p, st); else launch_conv<64, 64>(p, st); }
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_conv2d_grouped(const aod_conv_desc_t* desc, int ngroups, const void* const* src, const void* const* w_packed,
                                  void* const* dst, const float* const* pre_shift, const void* const* mask, float* const* colsum,
                                  aod_stream_t stream) {
  AOD_CHECK_ARG(desc && src && w_packed && dst && ngroups >= 1 && ngroups <= 4, "conv_grouped: 1..4 groups");
  AOD_CHECK_ARG(!desc->out_f32, "conv_grouped: bf16 destinations only");
  AOD_CHECK_ARG(!desc->x3 || (desc->N % 256 == 0 && desc->R * desc->S * desc->C >= 2048), "conv_grouped (x3): N %% 256 == 0 and a deep K required");
  ConvKParams p;
  int rc = conv_params(desc, src[0], w_packed[0], dst[0], nullptr, pre_shift ? pre_shift[0] : nullptr, nullptr, mask ? mask[0] : nullptr, nullptr,
                       nullptr, colsum ? colsum[0] : nullptr, p);
  if (rc) return rc;
  if (p.M == 0) return 0;
  AOD_CHECK_ARG(!p.perm, "conv_grouped: class-major stride-2 dgrad launches are not grouped");
  p.ngroups = ngroups;
  bool any_mask = false;
  for (int g = 0; g < ngroups; ++g) {
    AOD_CHECK_ARG(src[g] && w_packed[g] && dst[g], "conv_grouped: null operand in group %d", g);
    p.grp[g].x = (const bf16_t*)src[g]; p.grp[g].w = (const bf16_t*)w_packed[g]; p.grp[g].y = dst[g];
    p.grp[g].pre_shift = pre_shift ? pre_shift[g] : nullptr;
    p.grp[g].mask = mask ? (const bf16_t*)mask[g] : nullptr;
    p.grp[g].colsum = colsum ? colsum[g] : nullptr;
    any_mask = any_mask || p.grp[g].mask;
  }
  hipStream_t st = (hipStream_t)stream;
  // the tile whose grouped tile count fills the CU rounds best: 256 x 256 at one workgroup per CU, else 128 x 128 on 8 waves at two
  auto fill = [&](int bm, int bn, int slots) {
    const long long t = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * ngroups;
    return (double)t / (double)(((t + slots - 1) / slots) * slots);
  };
  static const char* dbg_t256 = getenv("AOD_TILE_256");
  const bool ok256 = !(dbg_t256 && dbg_t256[0] == '0') && p.N % 256 == 0 && p.K >= 1024;
  if (p.x3 && p.R == p.S && !p.bigrows) {
    // the head towers' grouped launches CAN run on the persistent producer / consumer kernel (conv_x3p.hip, 128 x 256 tiles; identical bits):
    // AOD_X3P_GROUPED=1.  Not the default -- launch by launch the two forms are level (671 vs 689 us forward, 729 vs 725 us dgrad for three
    // groups at 16 x 512^2, profiles/r06_x3p_micro.txt) and inside the step the 256 x 256 tile is 0.3 - 0.5 ms ahead
    const char* dbg_g = getenv("AOD_X3P_GROUPED");                   // (read per call: tests / A-B scripts switch it in-process)
    bool any_cs = false;
    for (int g = 0; g < ngroups; ++g) any_cs = any_cs || p.grp[g].colsum;
    if (dbg_g && dbg_g[0] == '1') {        // (opt-in: interleaved A/Bs put it level with the 256 x 256 tile, tools/dbg/x3p_grouped_micro.py)
      X3PArgs a;
      x3p_args(p, a);
      if (ngroups == 1) { a.ngroups = 1; a.grp[0].x = p.grp[0].x; a.grp[0].w = p.grp[0].w; a.grp[0].y = reinterpret_cast<bf16_t*>(p.grp[0].y);
                          a.grp[0].shift = p.grp[0].pre_shift; a.grp[0].mask = p.grp[0].mask; a.grp[0].colsum = p.grp[0].colsum; }
      if (aod_conv_x3p_wants(a, any_cs && aod_get_deterministic())) return aod_conv_x3p_launch(a, st);
    }
  }
  if (p.x3) {          // x3 groups take the big tile (the 4-wave forms have no grouped instances)
    if (!any_mask) launch_conv<256, 256, 512, 0, true, 2, true>(p, st); else launch_conv<256, 256, 512, 1, true, 2, true>(p, st);
  } else if (ok256 && fill(256, 256, 256) * 1.12 >= fill(128, 128, 512)) {        // (the big tile is ~15 % faster per FLOP when its rounds are full)
    if (!any_mask) launch_conv<256, 256, 512, 0, true>(p, st); else launch_conv<256, 256, 512, 1, true>(p, st);
  } else {
    AOD_CHECK_ARG(p.N >= 128, "conv_grouped: N >= 128 required");
    if (!any_mask) launch_conv<128, 128, 512, 0, true>(p, st); else launch_conv<128, 128, 512, 1, true>(p, st);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_conv2d(const aod_conv_desc_t* desc, const void* src, const void* w_packed, void* dst,
                          const float* pre_scale, const float* pre_shift, const void* res, const void* mask,
                          const float* post_scale, void* zraw, float* colsum, aod_stream_t stream) {
  return aod_conv2d_ws(desc, src, w_packed, dst, pre_scale, pre_shift, res, mask, post_scale, zraw, colsum, nullptr, 0, stream);
}

// =====================================================================================
// wgrad
// =====================================================================================
// Row table: one 32-B record per GEMM row m (= destination pixel): where its receptive field starts.
// Depends only on the segment geometry / stride / pad, so it is built once per geometry and shared by every conv
// (and every iteration) with that geometry -- the per-step im2col decode of wgrad becomes one 32-B record per pixel.
struct RowRec {       // 32 B: two 16-B pieces, fetched by LDS-DMA (one wave-instruction = the records of 32 pixels)
  unsigned xrow;     // byte offset of the top-left tap pixel (b, y0, x0) in the source (wraps for y0/x0 < 0: only used when valid)
  unsigned zoff;     // byte offset of this pixel's row in the dZ buffer
  unsigned wc2;      // source row pitch in bytes: W * C * 2
  unsigned pad_;
  unsigned long long mask;   // bit (r*S + s) set <=> tap (r, s) lies inside the image for this pixel (R*S <= 64)
  unsigned long long pad2_;
};
static_assert(sizeof(RowRec) == 32, "row record layout");

__global__ void row_table_kernel(const ConvKParams p, RowRec* __restrict__ tab) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  int sg = 0, mstart = 0;
  while (sg < p.nseg - 1 && m >= p.seg_mend[sg]) { mstart = p.seg_mend[sg]; ++sg; }
  const int ml = m - mstart;
  const int ohw = p.segOH[sg] * p.segOW[sg];
  const int b = ml / ohw, rem = ml - b * ohw;
  const int oy = rem / p.segOW[sg], ox = rem - oy * p.segOW[sg];
  const int H = p.segH[sg], W = p.segW[sg];
  const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
  RowRec r;
  r.xrow = (unsigned)((p.seg_src0[sg] + (long long)b * H * W + (long long)y0 * W + x0) * p.C * 2);
  r.zoff = (unsigned)((p.seg_dst0[sg] + ml) * (long long)p.N * 2);
  r.wc2 = (unsigned)(W * p.C * 2);
  unsigned long long mk = 0;
  for (int tr = 0; tr < p.R; ++tr)
    for (int ts = 0; ts < p.S; ++ts) {
      const int y = y0 + tr * p.dil, x = x0 + ts * p.dil;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) mk |= 1ull << (tr * p.S + ts);
    }
  r.mask = mk;
  r.pad_ = 0; r.pad2_ = 0;
  tab[m] = r;
}

extern "C" size_t aod_conv_row_table_bytes(const aod_conv_desc_t* d) {
  long long m = 0;
  for (int i = 0; i < d->nseg; ++i) m += (long long)d->seg[i].B * d->seg[i].OH * d->seg[i].OW;
  return (size_t)m * sizeof(RowRec);
}

extern "C" int aod_conv_row_table(const aod_conv_desc_t* d, void* table, aod_stream_t stream) {
  AOD_CHECK_ARG(d && table && !d->transposed, "row_table: bad args");
  ConvKParams cp;
  memset(&cp, 0, sizeof(cp));
  int rc = fill_params(d, cp);
  if (rc) return rc;
  if (cp.M == 0) return 0;
  AOD_CHECK_ARG(d->R * d->S <= 64, "row_table: at most 64 filter taps (got %d)", d->R * d->S);
  hipLaunchKernelGGL(row_table_kernel, dim3((cp.M + 255) / 256), dim3(256), 0, (hipStream_t)stream, cp, (RowRec*)table);
  AOD_LAUNCH_CHECK();
  return 0;
}

struct WgradParams {
  const bf16_t* x;
  const bf16_t* dz;
  float* dw;
  const RowRec* tab;
  int C, N, K, R, S, dil;
  int M;
  int tiles_n, tiles_k, splits, rows_per_split, stagger, xcd_order, x3;
  long long x_bytes, z_bytes, tab_bytes;
  long long slab_stride;     // > 0: split s STORES its partial tile into dw + s * slab_stride (deterministic, summed by the unpack); 0: fp32 atomics
};

// byte offset of (row, 16-B chunk) in a 256-B-pitch bf16 image that serves transposed reads
__device__ __forceinline__ int tr_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// dW[n][kk] += sum_m dZ[m][n] * X[pix(m, tap(kk))][c(kk)]: (128 TN) x (128 TK) output tile per workgroup, 64 pixels per step.
// Both operands are staged as 128-column SUB-IMAGES that keep the global row-major form ([pixel][128 columns], 256-B rows), TN of dZ and
// TK of X per stage, filled by LDS-DMA: one wave-instruction = 4 pixel rows x 16 chunks, the conflict-avoiding XOR applied on the SOURCE
// chunk index, which is the same for the 4 rows a lane serves -> every lane owns ONE fixed (tap, channel-chunk) column per sub-image.
// NW waves per workgroup (4 or 8).  128 x 128 tile: split 2 x 2 (64 x 64 per wave) or 2 x 4 (64 x 32 per wave), two workgroups per
// CU.  256 x 256 tile (TN = TK = 2, 8 waves as 2 x 4, 128 x 64 per wave, one workgroup per CU): the fragment reads are 8-B transposed
// reads, so the 128 x 128 form moves 768 B of LDS per MFMA -- more than the LDS delivers at the matrix pipe's rate; the big tile halves
// that (and the LDS-DMA traffic per MFMA), like the 256 x 256 tile of the forward kernel.
// X3 (aod_conv_desc_t.x3): dZ and X are X-layout rows, so a tile of dW' = dZ'^T X' holds the four products (zh, zl) x (xh, xl) of its
// logical entries in 32-wide bands.  A wave's block starts on a 64-column boundary in both directions and is a whole number of
// [h32 | l32] band pairs (4-wave 128 x 128 form: 64 x 64 per wave; 8-wave 256 x 256 form: 128 x 64): its zl x xl quarter (2^-16 of the
// result) is skipped at compile time, which leaves exactly the three products of the forward form; the unpack kernel adds the three bands
// of every slab.
template <int NW, int TN, int TK, bool X3>
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, int bid_in) {
  static_assert(!X3 || (NW == 4 && TN == 1 && TK == 1) || (NW == 8 && TN == 2 && TK == 2), "X3: wave blocks must be whole [h32 | l32] band pairs");
  constexpr int RPW = 4 * NW;          // pixel rows covered per pass of all waves
  constexpr int PASSES = 64 / RPW;
  constexpr int WNC = NW / 2;          // waves along the (tap, channel) axis
  constexpr int NI = 4 * TN;           // 16-row groups per wave along the dZ-channel axis (2 waves along it)
  constexpr int NJ = 8 * TK / WNC;     // 16-column groups per wave along the (tap, channel) axis
  constexpr int BKM = 64;
  constexpr int IMG = BKM * 256;       // one 128-column sub-image
  constexpr int STAGE = (TN + TK) * IMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int wm = uw / WNC, wn = uw % WNC;
  // bid_in runs (pixel split, tile) with the TILE fastest, over a range that one XCD executes contiguously (xcd_swizzle at the call site):
  // the tiles of a split -- the nine taps of a 3x3 filter read the same dZ rows and the same x rows at shifted positions -- share that XCD's
  // L2.  With the split fastest they were spread over all eight L2s and every operand row crossed HBM once per tile: 2.78 GB per grouped
  // tower launch for 0.36 GB of operands, i.e. the kernel ran at the HBM rate (profiles/r03_pmc_passes.txt).
  int bid = bid_in;
  const int ntile = p.tiles_n * p.tiles_k;
  const int split = bid / ntile; bid -= split * ntile;
  const int tile_k = bid % p.tiles_k, tile_n = bid / p.tiles_k;
  const int n0 = tile_n * (128 * TN), k0 = tile_k * (128 * TK);
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  if (ms >= me) return;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (int)p.z_bytes, 0x00020000);
  const int prow = lane >> 4;                                  // pixel row inside the wave's 4-row group
  const int ch = (lane & 15) ^ ((prow << 2) | (uw & 3));       // source chunk of this lane (fixed)
  constexpr unsigned OOB_BASE = 0xf0000000u;

  // per-lane constants of each sub-image: dZ column, and for X the byte offset of the lane's (tap, channel chunk) relative to a pixel's
  // top-left tap, its row offset and its mask bit
  unsigned zcol[TN];
  bool zok[TN];
#pragma unroll
  for (int h = 0; h < TN; ++h) {
    const int zn = n0 + 128 * h + ch * 8;
    zok[h] = zn < p.N;
    zcol[h] = (unsigned)(zn * 2);
  }
  unsigned cdx[TK];
  int dy[TK];
  unsigned long long tbit[TK];
#pragma unroll
  for (int h = 0; h < TK; ++h) {
    const int kk = k0 + 128 * h + ch * 8;
    const bool kok = kk < p.K;
    const int tap = kok ? kk / p.C : 0, c0 = kok ? kk - tap * p.C : 0;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    dy[h] = tr * p.dil;
    cdx[h] = (unsigned)((ts * p.dil * p.C + c0) * 2);
    tbit[h] = kok ? (1ull << tap) : 0ull;
  }
  // Row records reach the lanes through LDS: waves 0 and 1 fetch the 64 records of a step with ONE LDS-DMA instruction each (two steps
  // ahead, into a two-slot ring behind the operand stages), every lane then picks its four with ds_read.  Fetching them into VGPRs
  // (global_load) beside the LDS-DMA operand loads made every K-step drain the whole vector-memory queue: 27 % of the kernel.
  const auto rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.tab, 0, (int)p.tab_bytes, 0x00020000);
  char* const stab = smem + 2 * STAGE;                        // [2][64] records
  auto tdma = [&](int mbase, int slot) {
    if (uw < 2) {
      const int m = mbase + 32 * uw + (lane >> 1);
      const unsigned off = (unsigned)min(m, p.M - 1) * 32u + (unsigned)(lane & 1) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (__attribute__((address_space(3))) void*)(stab + slot * 2048 + uw * 1024), 16, off, 0, 0, 0);
    }
  };
  RowRec rec[PASSES];
  auto rread = [&](int slot) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) rec[i] = *reinterpret_cast<const RowRec*>(stab + slot * 2048 + (RPW * i + 4 * uw + prow) * 32);
  };
  auto gload = [&](int mbase, int buf) {
    char* sz = smem + buf * STAGE;
    char* sx = sz + TN * IMG;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int m = mbase + RPW * i + 4 * uw + prow;
      const bool mok = m < me;
      const RowRec r = rec[i];
#pragma unroll
      for (int h = 0; h < TN; ++h) {
        const unsigned zoff = (mok && zok[h]) ? r.zoff + zcol[h] : OOB_BASE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_z, (__attribute__((address_space(3))) void*)(sz + h * IMG + (RPW * i + 4 * uw) * 256), 16, zoff, 0, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < TK; ++h) {
        const unsigned xoff = (mok && (r.mask & tbit[h])) ? r.xrow + (unsigned)dy[h] * r.wc2 + cdx[h] : OOB_BASE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sx + h * IMG + (RPW * i + 4 * uw) * 256), 16, xoff, 0, 0, 0);
      }
    }
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = (me - ms + BKM - 1) / BKM;
  tdma(ms, 0);
  if (nsteps > 1) tdma(ms + BKM, 1);
  __syncthreads();                                            // records of steps 0 and 1 are resident
  rread(0);
  gload(ms, 0);
  __syncthreads();
  // transposed-read lane roles: group g = lane>>4 covers k rows 8g..8g+7 of a 32-row sub-step; lane 4q+pp -> row q, cols 4pp..4pp+3
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  // stagger (8-wave form): waves 4-7 run half a step behind waves 0-3, with which they share their SIMDs (see conv_igemm_kernel)
  const bool late = p.stagger && NW == 8 && uw >= 4;
  bool carried = false;
  // Fragment reads are ISSUED from inline asm: behind the ds_read_tr builtin hipcc (ROCm 7.2) waits vmcnt(0) ahead of the first read of
  // every step -- i.e. for the LDS-DMA of the NEXT stage issued just above -- which serialises load and compute inside each wave (the
  // plain ds_read_b128 of the forward kernel does not get that wait).  The asm reads are invisible to the compiler's counters; frag_wait()
  // is their s_waitcnt and makes every destination opaque, so no MFMA is scheduled above it.
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  u32x2_t alo[NI], ahi[NI], blo[NJ], bhi[NJ];
  auto tr_read = [&](u32x2_t& dst, const char* ptr) {
    const unsigned a = (unsigned)(unsigned long long)ptr;       // (the low half of a shared-aperture address is the LDS byte address)
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(a) : "memory");
  };
  auto frag_read = [&](const char* sz, const char* sx, int ks) {
    const int r0 = ks * 32 + 8 * g + q;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = wm * (16 * NI) + i * 16 + pp * 4;       // column of the dZ stage: sub-image col >> 7
      const char* im = sz + (col >> 7) * IMG;
      const int c = col & 127;
      tr_read(alo[i], im + tr_off(r0, c >> 3) + (c & 7) * 2);
      tr_read(ahi[i], im + tr_off(r0 + 4, c >> 3) + (c & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wn * (16 * NJ) + j * 16 + pp * 4;
      const char* im = sx + (col >> 7) * IMG;
      const int c = col & 127;
      tr_read(blo[j], im + tr_off(r0, c >> 3) + (c & 7) * 2);
      tr_read(bhi[j], im + tr_off(r0 + 4, c >> 3) + (c & 7) * 2);
    }
  };
  bf16x8 af[NI], bfr[NJ];
  auto frag_wait = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < NI; ++i) af[i] = __builtin_bit_cast(bf16x8, (u32x4_t){alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]});
#pragma unroll
    for (int j = 0; j < NJ; ++j) bfr[j] = __builtin_bit_cast(bf16x8, (u32x4_t){blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]});
  };
  auto mfma_block = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (X3 && (i & 3) >= 2 && (j & 3) >= 2) continue;       // tail x tail (16-row groups 2, 3 of every four = a tail band)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
  };
  for (int stp = 0; stp < nsteps; ++stp) {
    const int cur = stp & 1;
    if (late && carried) mfma_block();
    if (stp + 1 < nsteps) {
      rread((stp + 1) & 1);                                 // landed before the previous barrier
      gload(ms + (stp + 1) * BKM, cur ^ 1);
    }
    // slot stp & 1 held this step's records; every wave read them one iteration ago (ahead of the barrier), so it can be refilled
    if (stp + 2 < nsteps) tdma(ms + (stp + 2) * BKM, stp & 1);
    const char* sz = smem + cur * STAGE;
    const char* sx = sz + TN * IMG;
    frag_read(sz, sx, 0);
    frag_wait();
    mfma_block();
    frag_read(sz, sx, 1);
    frag_wait();                                            // (also in the late waves: their reads of this stage end before the barrier)
    if (!late) mfma_block(); else carried = true;
    __syncthreads();
  }
  if (late && carried) mfma_block();
  // slab mode: every pixel split owns a slab and STORES its partial tile (plain dword stores run at ~6 TB/s chip-wide, float atomics at
  // ~1.3 TB/s: 512 workgroups x 64 KB of atomics were a 25 us tail on every launch); the unpack kernel adds the slabs in split order,
  // so the weight gradient no longer depends on arrival order.  Otherwise: accumulate into dW[n][kk] with fp32 atomics (one dword per
  // lane, 16 consecutive columns per row group).
  float* const dwp = p.dw + (long long)split * p.slab_stride;
  const bool slab = p.slab_stride > 0;
  const int lr = lane & 15, lq = lane >> 4;
  if constexpr (X3) {
    // X3: the three bands of a logical entry sit in the SAME lane and register slot of three accumulators of this wave -- (zh, xh), (zh, xl)
    // and (zl, xh) -- and are added here, (hh + hl) + lh: the slab is the LOGICAL [N / 2][R][S][C / 2] partial gradient (a quarter of the
    // bytes of the band form) and the plain unpack kernel serves it
    const int KL = p.K >> 1;
#pragma unroll
    for (int I = 0; I < NI / 2; ++I)
#pragma unroll
      for (int J = 0; J < NJ / 2; ++J) {
        const int ih = (I >> 1) * 4 + (I & 1), jh = (J >> 1) * 4 + (J & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wm * (16 * NI) + ih * 16 + lq * 4 + r;        // physical column of the head of the entry's dZ channel
          const int k = k0 + wn * (16 * NJ) + jh * 16 + lr;                // ... and of its (tap, input channel)
          if (n < p.N && k < p.K) {
            const float v = (acc[ih][jh][r] + acc[ih][jh + 2][r]) + acc[ih + 2][jh][r];
            dwp[(long long)(((n >> 6) << 5) + (n & 31)) * KL + ((k >> 6) << 5) + (k & 31)] = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * (16 * NI) + i * 16 + lq * 4 + r;
        const int k = k0 + wn * (16 * NJ) + j * 16 + lr;
#ifdef AOD_WGRAD_NO_EPI      // ablation build (tools/dbg): how much of the kernel is the atomic epilogue
        if (n < p.N && k < p.K && acc[i][j][r] == 12345.678f) p.dw[(long long)n * p.K + k] = 1.f;
#else
        if (n < p.N && k < p.K) {
          if (slab) dwp[(long long)n * p.K + k] = acc[i][j][r];
          else atomicAdd(p.dw + (long long)n * p.K + k, acc[i][j][r]);
        }
#endif
      }
}

// X3, WIDE form (the tower filters and every other layer whose X-layout dW is whole 512 x 512 tiles): a 256 x 256 LOGICAL tile per
// workgroup with ONE accumulator per entry -- zh*xh + zh*xl + zl*xh are summed in it, like the forward form sums its three products.  The
// band form above spends a 256 x 256 accumulator tile on 128 x 128 logical entries (a quarter of it on the tail x tail band it never
// computes) and issues 1.5 MFMAs per accumulator and 64-pixel step for 64 KB of staged operands; this form stages the same 64 KB per
// 32-pixel step -- eight 128-column sub-images: [zh32 | zl32] x 4 channel groups for each operand -- for 3 MFMAs per accumulator: twice the
// matrix work per staged byte, four times per LDS fragment read.  8 waves as 2 x 4, 128 x 64 logical entries per wave (32 accumulators);
// the head fragments of dZ are replaced by its tail fragments for the third product (the x fragments stay).
// NW = 8, T = 4: 512 x 512 physical columns = 256 x 256 logical entries, one workgroup per CU.  NW = 4, T = 2: 256 x 256 physical = 128 x 128
// logical (2 x 2 waves, 64 x 64 logical per wave, 16 accumulators; two workgroups per CU): the backbone layers whose dW is not whole
// 512-column tiles or whose pixel axis is too short to fill the chip with the big form.
template <int NW, int T>
__device__ __forceinline__ void wgrad_tile_x3w(const WgradParams& p, int bid_in) {
  static_assert((NW == 8 && T == 4) || (NW == 4 && T == 2), "wide x3 wgrad forms");
  constexpr int TN = T, TK = T, BKM = 32;
  constexpr int RPW = 4 * NW, PASSES = BKM / RPW;      // pixel rows per pass of all waves
  constexpr int IMG = BKM * 256;              // one 128-column sub-image: 8 KB
  constexpr int STAGE = (TN + TK) * IMG;      // 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  constexpr int WNC = NW / 2;                           // waves along the (tap, channel) axis
  const int wm = uw / WNC, wn = uw % WNC;
  int bid = bid_in;
  const int ntile = p.tiles_n * p.tiles_k;
  const int split = bid / ntile; bid -= split * ntile;
  const int tile_k = bid % p.tiles_k, tile_n = bid / p.tiles_k;
  const int n0 = tile_n * (128 * T), k0 = tile_k * (128 * T);       // physical columns
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  if (ms >= me) return;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (int)p.z_bytes, 0x00020000);
  const auto rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.tab, 0, (int)p.tab_bytes, 0x00020000);
  const int prow = lane >> 4;                                  // pixel row inside the wave's 4-row group
  const int ch = (lane & 15) ^ ((prow << 2) | (uw & 3));       // source chunk of this lane (fixed; tr_off's key of row RPW * i + 4 * uw + prow)
  constexpr unsigned OOB_BASE = 0xf0000000u;
  unsigned zcol[TN];
  bool zok[TN];
#pragma unroll
  for (int h = 0; h < TN; ++h) {
    const int zn = n0 + 128 * h + ch * 8;
    zok[h] = zn < p.N;
    zcol[h] = (unsigned)(zn * 2);
  }
  unsigned cdx[TK];
  int dy[TK];
  unsigned long long tbit[TK];
#pragma unroll
  for (int h = 0; h < TK; ++h) {
    const int kk = k0 + 128 * h + ch * 8;
    const bool kok = kk < p.K;
    const int tap = kok ? kk / p.C : 0, c0 = kok ? kk - tap * p.C : 0;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    dy[h] = tr * p.dil;
    cdx[h] = (unsigned)((ts * p.dil * p.C + c0) * 2);
    tbit[h] = kok ? (1ull << tap) : 0ull;
  }
  char* const stab = smem + 2 * STAGE;                        // [2][32] row records
  auto tdma = [&](int mbase, int slot) {
    if (uw == 0) {
      const int m = mbase + (lane >> 1);
      const unsigned off = (unsigned)min(m, p.M - 1) * 32u + (unsigned)(lane & 1) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (__attribute__((address_space(3))) void*)(stab + slot * 1024), 16, off, 0, 0, 0);
    }
  };
  RowRec rec[PASSES];
  auto rread = [&](int slot) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) rec[i] = *reinterpret_cast<const RowRec*>(stab + slot * 1024 + (RPW * i + 4 * uw + prow) * 32);
  };
  auto gload = [&](int mbase, int buf) {
    char* sz = smem + buf * STAGE;
    char* sx = sz + TN * IMG;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int m = mbase + RPW * i + 4 * uw + prow;
      const bool mok = m < me;
      const RowRec r = rec[i];
#pragma unroll
      for (int h = 0; h < TN; ++h) {
        const unsigned zoff = (mok && zok[h]) ? r.zoff + zcol[h] : OOB_BASE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_z, (__attribute__((address_space(3))) void*)(sz + h * IMG + (RPW * i + 4 * uw) * 256), 16, zoff, 0, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < TK; ++h) {
        const unsigned xoff = (mok && (r.mask & tbit[h])) ? r.xrow + (unsigned)dy[h] * r.wc2 + cdx[h] : OOB_BASE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sx + h * IMG + (RPW * i + 4 * uw) * 256), 16, xoff, 0, 0, 0);
      }
    }
  };
  // 16-entry groups per wave: NW = 8: 128 logical dZ channels x 64 logical (tap, channel) columns; NW = 4: 64 x 64
  constexpr int NI = 32 * T / 16, NJ = 4;
  constexpr int WML = 16 * NI, WNL = 16 * NJ;
  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nsteps = (me - ms + BKM - 1) / BKM;
  tdma(ms, 0);
  if (nsteps > 1) tdma(ms + BKM, 1);
  __syncthreads();
  rread(0);
  gload(ms, 0);
  __syncthreads();
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto tr_read = [&](u32x2_t& dst, const char* ptr) {
    const unsigned a = (unsigned)(unsigned long long)ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(a) : "memory");
  };
  // fragment of the 16 physical columns [col, col + 16) of an operand stage: pixel rows 8g + q (+4), 4 columns per lane
  auto frag = [&](u32x2_t& lo, u32x2_t& hi, const char* base, int col) {
    const char* im = base + (col >> 7) * IMG;
    const int c = (col & 127) + pp * 4;
    tr_read(lo, im + tr_off(8 * g + q, c >> 3) + (c & 7) * 2);
    tr_read(hi, im + tr_off(8 * g + q + 4, c >> 3) + (c & 7) * 2);
  };
  u32x2_t alo[NI], ahi[NI], bhlo[NJ], bhhi[NJ], bllo[NJ], blhi[NJ];
  bf16x8 af[NI], bh[NJ], bl[NJ];
  for (int stp = 0; stp < nsteps; ++stp) {
    const int cur = stp & 1;
    if (stp + 1 < nsteps) {
      rread((stp + 1) & 1);                                 // landed before the previous barrier
      gload(ms + (stp + 1) * BKM, cur ^ 1);
    }
    if (stp + 2 < nsteps) tdma(ms + (stp + 2) * BKM, stp & 1);
    const char* sz = smem + cur * STAGE;
    const char* sx = sz + TN * IMG;
    // heads of dZ, heads and tails of x
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int nl = wm * WML + i * 16; frag(alo[i], ahi[i], sz, ((nl >> 5) << 6) + (nl & 31)); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int kl = wn * WNL + j * 16, col = ((kl >> 5) << 6) + (kl & 31);
      frag(bhlo[j], bhhi[j], sx, col);
      frag(bllo[j], blhi[j], sx, col + 32);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) { asm volatile("" : "+v"(alo[i]), "+v"(ahi[i])); af[i] = __builtin_bit_cast(bf16x8, (u32x4_t){alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]}); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      asm volatile("" : "+v"(bhlo[j]), "+v"(bhhi[j]), "+v"(bllo[j]), "+v"(blhi[j]));
      bh[j] = __builtin_bit_cast(bf16x8, (u32x4_t){bhlo[j][0], bhlo[j][1], bhhi[j][0], bhhi[j][1]});
      bl[j] = __builtin_bit_cast(bf16x8, (u32x4_t){bllo[j][0], bllo[j][1], blhi[j][0], blhi[j][1]});
    }
    // tails of dZ requested now, consumed after the two head products
    u32x2_t tlo[NI], thi[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int nl = wm * WML + i * 16; frag(tlo[i], thi[i], sz, ((nl >> 5) << 6) + (nl & 31) + 32); }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bh[j], acc[i][j], 0, 0, 0);       // zh * xh
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);       // zh * xl
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) { asm volatile("" : "+v"(tlo[i]), "+v"(thi[i])); af[i] = __builtin_bit_cast(bf16x8, (u32x4_t){tlo[i][0], tlo[i][1], thi[i][0], thi[i][1]}); }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bh[j], acc[i][j], 0, 0, 0);         // zl * xh
    __syncthreads();
  }
  // logical slab [N / 2][K / 2] of this pixel split
  float* const dwp = p.dw + (long long)split * p.slab_stride;
  const int lr = lane & 15, lq = lane >> 4;
  const int NL = p.N >> 1, KL = p.K >> 1;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (n0 >> 1) + wm * WML + i * 16 + lq * 4 + r;
        const int k = (k0 >> 1) + wn * WNL + j * 16 + lr;
        if (n < NL && k < KL) dwp[(long long)n * KL + k] = acc[i][j][r];
      }
}
template <int NW, int T>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_x3w_kernel(const WgradParams p) {
  wgrad_tile_x3w<NW, T>(p, p.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x);
}

// split of the pixel axis over workgroups: all workgroups co-resident (<= 2 per CU, no ragged second round); cost model (measured on
// MI355X): one 64-pixel step costs ~1.45 us with one workgroup per CU and ~1.7 us with two (both share the CU); every workgroup ends with
// 64 KB of output -- fp32 atomics at ~1.3 TB/s chip-wide (0.05 us per workgroup) or, in slab mode, plain stores at ~5.5 TB/s (0.012 us)
// plus the unpack kernel's read of one more slab per split.
template <int NW, int TN, int TK, bool X3 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(TN * TK > 1 ? 2 : NW / 2, TN * TK > 1 ? 2 : NW / 2))) void conv_wgrad_kernel(const WgradParams p) {
  wgrad_tile<NW, TN, TK, X3>(p, p.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x);
}
// Grouped launch: up to WG_MAXG weight gradients (ANY geometries, one tile form) share one grid.  Alone a backbone layer needs 100+ pixel
// splits of its few 128 x 128 tiles to fill the chip -- a 512 x 128 filter (256 KB) leaves 30 MB of partial slabs for the unpack; three
// layers together need a third of the splits each (longer pixel runs per workgroup, a third of the slab traffic, one launch).
constexpr int WG_MAXG = 4;
struct WgradGroups { WgradParams g[WG_MAXG]; int wg0[WG_MAXG + 1]; int n; };
template <int NW, int TN, int TK, bool X3 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(TN * TK > 1 ? 2 : NW / 2, TN * TK > 1 ? 2 : NW / 2))) void conv_wgrad_grouped_kernel(const WgradGroups gp) {
  const int b = gp.g[0].xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  // (selects, not an indexed read of the argument block: a run-time index would move the whole array to scratch memory)
  if (b < gp.wg0[1]) wgrad_tile<NW, TN, TK, X3>(gp.g[0], b);
  else if (b < gp.wg0[2]) wgrad_tile<NW, TN, TK, X3>(gp.g[1], b - gp.wg0[1]);
  else if (b < gp.wg0[3]) wgrad_tile<NW, TN, TK, X3>(gp.g[2], b - gp.wg0[2]);
  else wgrad_tile<NW, TN, TK, X3>(gp.g[3], b - gp.wg0[3]);
}

template <int NW, int T>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_grouped_x3w_kernel(const WgradGroups gp) {
  const int b = gp.g[0].xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  if (b < gp.wg0[1]) wgrad_tile_x3w<NW, T>(gp.g[0], b);
  else if (b < gp.wg0[2]) wgrad_tile_x3w<NW, T>(gp.g[1], b - gp.wg0[1]);
  else if (b < gp.wg0[3]) wgrad_tile_x3w<NW, T>(gp.g[2], b - gp.wg0[2]);
  else wgrad_tile_x3w<NW, T>(gp.g[3], b - gp.wg0[3]);
}

// `big`: tile form -- 0 = 128 x 128, 1 = 256 x 256, 2 = the WIDE x3 form of 8 waves (512 x 512 physical columns = 256 x 256 logical entries,
// wgrad_tile_x3w<8, 4>), 3 = the wide x3 form of 4 waves (256 x 256 physical = 128 x 128 logical, wgrad_tile_x3w<4, 2>)
static void wgrad_plan(int M, int N, int K, bool slabs, int& tiles_n, int& tiles_k, int& splits, int& rps, int& big, int big_min_m = 49152, int x3 = 0) {
  // the 256 x 256 tile (one workgroup per CU): deep layers whose dW is whole tiles of it and whose pixel axis gives every CU a long run
  static const char* dbg_big = getenv("AOD_WGRAD_256");
  // (measured, single launches: 16 x 32 x 32 pixels lose 15 % with the big tile, 16 x 64 x 64 gain 10 %; members of a GROUP get longer pixel
  // runs per workgroup and take it from 16 384 pixels on: -0.11 ms per step, AOD_WGRAD_BIG_MINM overrides the group threshold)
  big = (N % 256 == 0 && K % 256 == 0 && M >= big_min_m) ? 1 : 0;
  if (dbg_big && dbg_big[0] == '0') big = 0;
  static const char* dbg_wide = getenv("AOD_WGRAD_X3_WIDE");       // (debug: 0 = never the wide x3 form)
  if (x3 && !(dbg_wide && dbg_wide[0] == '0')) {
    if (N % 512 == 0 && K % 512 == 0 && M >= big_min_m && !(dbg_big && dbg_big[0] == '0')) big = 2;
    else if (N % 256 == 0 && K % 256 == 0) big = 3;
    else {
      // narrow dZ (the prediction convs: 384 / 128 / 64 physical columns) on a long pixel axis: the wide 4-wave form with a ragged column
      // tile (its loads of columns >= N return zeros, its stores are bounds-checked) -- 128 FLOP per staged byte instead of 64, which is what
      // bounds the 128 x 128 form.  AOD_WGRAD_X3_RAGGED=0: the 128 x 128 form.
      static const char* dbg_rag = getenv("AOD_WGRAD_X3_RAGGED");
      // (64 columns -- retina_L -- stay on the 128 x 128 form: 118 vs 137 us; the 8-wave wide form for the 384-column retina_cls measured the
      // same as this one, profiles/r05_x3_tile_ab.txt)
      if (!(dbg_rag && dbg_rag[0] == '0') && N % 64 == 0 && N >= 128 && K % 256 == 0 && M >= 16384) big = 3;
    }
  }
  const int T = big == 2 ? 512 : (big ? 256 : 128);
  tiles_n = (N + T - 1) / T;
  tiles_k = (K + T - 1) / T;
  const int tiles = tiles_n * tiles_k;
  int best = 1;
  double best_cost = 1e30;
  const int slots = (big == 1 || big == 2) ? 256 : 512;
  const int max_s = slots / tiles > 0 ? slots / tiles : 1;
  for (int sp = 1; sp <= max_s; ++sp) {
    const int rows = ((M + sp - 1) / sp + 63) / 64 * 64;
    const int nsp = (M + rows - 1) / rows;
    const int wgs = tiles * nsp;
    double cost = (rows / 64) * (big == 2 ? 4.5 : (big == 3 ? (wgs > 256 ? 2.6 : 2.2) : (big ? 2.9 : (wgs > 256 ? 1.7 : 1.45))));
    cost += slabs ? wgs * 0.012 * (big ? 4 : 1) + nsp * ((double)N * K * (x3 ? 1.0 : 4.0)) / 2.5e6 : wgs * 0.05 * (big ? 4 : 1);
    if (cost < best_cost) { best_cost = cost; best = sp; }
  }
  rps = ((M + best - 1) / best + 63) / 64 * 64;
  splits = (M + rps - 1) / rps;
}

// operands + geometry of one weight gradient (everything but the split plan)
static int wgrad_fill(const aod_conv_desc_t* d, const void* x, const void* dz, float* dw, const void* row_table, WgradParams& p) {
  AOD_CHECK_ARG(d && x && dz && dw && row_table, "wgrad: null pointer");
  AOD_CHECK_ARG(!d->transposed, "wgrad: descriptor must be the forward descriptor");
  AOD_CHECK_ARG(d->N % 8 == 0, "wgrad: N %d must be a multiple of 8 (pad dZ)", d->N);
  ConvKParams cp;
  memset(&cp, 0, sizeof(cp));
  int rc = fill_params(d, cp);
  if (rc) return rc;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.dz = (const bf16_t*)dz; p.dw = dw; p.tab = (const RowRec*)row_table;
  p.C = cp.C; p.N = cp.N; p.K = cp.K; p.R = cp.R; p.S = cp.S; p.dil = cp.dil; p.M = cp.M; p.x3 = cp.x3;
  if (p.x3) AOD_CHECK_ARG(p.N % 64 == 0 && p.C % 64 == 0, "wgrad (x3): X-layout widths (dZ %d, x %d) must be multiples of 64", p.N, p.C);
  long long xrows = 0, zrows = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const aod_conv_seg_t& sg = d->seg[i];
    AOD_CHECK_ARG(sg.src_row0 >= 0 && sg.dst_row0 >= 0, "wgrad: negative row offset");
    const long long e = sg.src_row0 + (long long)sg.B * sg.H * sg.W, ez = sg.dst_row0 + (long long)sg.B * sg.OH * sg.OW;
    if (e > xrows) xrows = e;
    if (ez > zrows) zrows = ez;
  }
  p.x_bytes = xrows * p.C * 2;
  p.z_bytes = zrows * p.N * 2;
  p.tab_bytes = (long long)p.M * (long long)sizeof(RowRec);
  AOD_CHECK_ARG(p.x_bytes < 0xe0000000ll && p.z_bytes < 0xe0000000ll, "wgrad: operand larger than 3.5 GiB (32-bit buffer offsets)");
  { static const char* dbg_st = getenv("AOD_STAGGER"); p.stagger = (dbg_st && dbg_st[0] == '0') ? 0 : 1; }
  { static const char* dbg_x = getenv("AOD_WGRAD_XCD"); p.xcd_order = (dbg_x && dbg_x[0] == '0') ? 0 : 1; }      // (debug: 0 = launch order)
  return 0;
}

static void wgrad_attrs() {
  static unsigned long long attr_done = 0;
  if (!aod_first_on_device(&attr_done)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<4, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<8, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<8, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_kernel<8, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_kernel<8, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<4, 1, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_kernel<4, 1, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<8, 2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_kernel<8, 2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x3w_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_x3w_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x3w_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_grouped_x3w_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
}

static int wgrad_launch(const aod_conv_desc_t* d, const void* x, const void* dz, float* dw, long long slab_stride, int max_slabs,
                        const void* row_table, aod_stream_t stream) {
  WgradParams p;
  int rc = wgrad_fill(d, x, dz, dw, row_table, p);
  if (rc) return rc;
  if (p.M == 0) return 0;
  int splits, rps, big;
  wgrad_plan(p.M, p.N, p.K, slab_stride > 0, p.tiles_n, p.tiles_k, splits, rps, big, 49152, p.x3);
  const int tiles = p.tiles_n * p.tiles_k;
  p.splits = splits; p.rows_per_split = rps;
  p.slab_stride = slab_stride;
  if (slab_stride > 0) {
    AOD_CHECK_ARG(splits <= max_slabs, "wgrad: %d slabs needed, %d provided (aod_conv2d_wgrad_splits)", splits, max_slabs);
    AOD_CHECK_ARG(slab_stride >= (long long)p.N * p.K / (p.x3 ? 4 : 1), "wgrad: slab stride smaller than N*K (x3: the logical N/2 * K/2)");
  }
  wgrad_attrs();
  static const char* dbg_w8 = getenv("AOD_WGRAD_W8");      // (debug: 0 = the 4-wave form)
  if (p.x3) {
    AOD_CHECK_ARG(slab_stride > 0, "wgrad (x3): slab form only");
    if (big == 2) hipLaunchKernelGGL((conv_wgrad_x3w_kernel<8, 4>), dim3(tiles * splits), dim3(512), 131072 + 4096, (hipStream_t)stream, p);
    else if (big == 3) hipLaunchKernelGGL((conv_wgrad_x3w_kernel<4, 2>), dim3(tiles * splits), dim3(256), 65536 + 4096, (hipStream_t)stream, p);
    else if (big) hipLaunchKernelGGL((conv_wgrad_kernel<8, 2, 2, true>), dim3(tiles * splits), dim3(512), 131072 + 4096, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<4, 1, 1, true>), dim3(tiles * splits), dim3(256), 65536 + 4096, (hipStream_t)stream, p);
  } else if (big) hipLaunchKernelGGL((conv_wgrad_kernel<8, 2, 2>), dim3(tiles * splits), dim3(512), 131072 + 4096, (hipStream_t)stream, p);
  else if (!(dbg_w8 && dbg_w8[0] == '0')) hipLaunchKernelGGL((conv_wgrad_kernel<8, 1, 1>), dim3(tiles * splits), dim3(512), 65536 + 4096, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((conv_wgrad_kernel<4, 1, 1>), dim3(tiles * splits), dim3(256), 65536 + 4096, (hipStream_t)stream, p);
  AOD_LAUNCH_CHECK();
  return 0;
}

// Split plan of a GROUP (slab form): every workgroup of the launch runs the same number T of 64-pixel steps (the groups' tiles are the same
// size, so that equalises their durations); T = the smallest for which the grid fits the chip's slots -- group g then gets ceil(steps_g / T)
// splits.  All members must take the same tile form (`big` of wgrad_plan); returns that form, or -1 when they do not.
static int wgrad_plan_group(int n, const int* M, const int* N, const int* K, int* tiles_n, int* tiles_k, int* splits, int* rps, int x3 = 0) {
  int big = -1;
  long long tiles[WG_MAXG];
  int steps[WG_MAXG], max_steps = 1;
  for (int g = 0; g < n; ++g) {
    int sp, r, b;
    static const char* dbg_minm = getenv("AOD_WGRAD_BIG_MINM");
    wgrad_plan(M[g], N[g], K[g], true, tiles_n[g], tiles_k[g], sp, r, b, n > 1 ? (dbg_minm ? atoi(dbg_minm) : 16384) : 49152, x3);
    if (big >= 0 && b != big) return -1;
    big = b;
    tiles[g] = (long long)tiles_n[g] * tiles_k[g];
    steps[g] = (M[g] + 63) / 64;
    if (steps[g] > max_steps) max_steps = steps[g];
  }
  static const char* dbg_slots = getenv("AOD_WGRAD_SLOTS");      // (debug: grid size the group plan aims at, small-tile form)
  const int slots = (big == 1 || big == 2) ? 256 : (dbg_slots ? atoi(dbg_slots) : 512);
  int T = 1;
  bool fits = false;
  for (; T <= max_steps; ++T) {
    long long tot = 0;
    for (int g = 0; g < n; ++g) tot += tiles[g] * ((steps[g] + T - 1) / T);
    if (tot <= slots) { fits = true; break; }
  }
  // more tiles than workgroup slots even with ONE split each (the X-layout tiles of the x3 mode are four times as many): a group would run
  // as several rounds of workgroups that each walk the whole pixel axis -- its members are better off alone, with their own splits
  if (!fits) { if (n > 1) return -1; T = max_steps; }
  if (x3 && n > 1) {
    // x3: a member often has so many tiles that the group gets one or two splits and fills the slots badly (4 tower filters = 144 big tiles:
    // one split each, 144 of 256 workgroup slots, 1 364 steps; alone each: 36 tiles x 7 splits = 252 slots, 195 steps).  Estimated duration
    // = rounds of the slots x steps per workgroup; group only if that does not lose against the members launched one by one.
    auto cost = [&](int g0, int g1) {
      long long best = -1;
      for (int t = 1; t <= max_steps; ++t) {
        long long tot = 0;
        for (int g = g0; g < g1; ++g) tot += tiles[g] * ((steps[g] + t - 1) / t);
        const long long c = ((tot + slots - 1) / slots) * t;
        if (best < 0 || c < best) best = c;
        if (tot <= slots / 2) break;         // (fewer workgroups than half the slots from here on: longer t only costs)
      }
      return best;
    };
    long long alone = 0;
    for (int g = 0; g < n; ++g) alone += cost(g, g + 1);
    const long long grouped = (long long)T;    // (one round by construction)
    if (grouped * 10 > alone * 11) return -1;
  }
  for (int g = 0; g < n; ++g) {
    rps[g] = T * 64;
    splits[g] = (M[g] + rps[g] - 1) / rps[g];
  }
  return big;
}

extern "C" int aod_conv2d_wgrad_group_plan(const aod_conv_desc_t* const* descs, int n, int32_t* splits_out) {
  AOD_CHECK_ARG(descs && splits_out && n >= 1 && n <= WG_MAXG, "wgrad_group_plan: 1..4 descriptors");
  int M[WG_MAXG], N[WG_MAXG], K[WG_MAXG], tn[WG_MAXG], tk[WG_MAXG], sp[WG_MAXG], rps[WG_MAXG];
  for (int g = 0; g < n; ++g) {
    AOD_CHECK_ARG(descs[g] && !descs[g]->transposed, "wgrad_group_plan: forward descriptors");
    ConvKParams cp;
    memset(&cp, 0, sizeof(cp));
    int rc = fill_params(descs[g], cp);
    if (rc) return rc;
    AOD_CHECK_ARG(cp.M > 0, "wgrad_group_plan: empty member");
    M[g] = cp.M; N[g] = cp.N; K[g] = cp.K;
    if (descs[g]->x3 != descs[0]->x3) return 1;      // (mixed precision modes never share a grid)
  }
  const int big = wgrad_plan_group(n, M, N, K, tn, tk, sp, rps, descs[0]->x3);
  if (big < 0) return 1;                       // mixed tile forms: launch the members one by one
  for (int g = 0; g < n; ++g) splits_out[g] = sp[g];
  return 0;
}

extern "C" int aod_conv2d_wgrad_grouped(const aod_conv_desc_t* const* descs, int n, const void* const* x, const void* const* dz,
                                        float* const* slabs, const int32_t* nslabs, const int64_t* slab_stride,
                                        const void* const* row_table, aod_stream_t stream) {
  AOD_CHECK_ARG(descs && x && dz && slabs && nslabs && slab_stride && row_table && n >= 1 && n <= WG_MAXG, "wgrad_grouped: 1..4 members");
  WgradGroups gp;
  memset(&gp, 0, sizeof(gp));
  int M[WG_MAXG], N[WG_MAXG], K[WG_MAXG], tn[WG_MAXG], tk[WG_MAXG], sp[WG_MAXG], rps[WG_MAXG];
  for (int g = 0; g < n; ++g) {
    int rc = wgrad_fill(descs[g], x[g], dz[g], slabs[g], row_table[g], gp.g[g]);
    if (rc) return rc;
    AOD_CHECK_ARG(gp.g[g].M > 0, "wgrad_grouped: empty member");
    M[g] = gp.g[g].M; N[g] = gp.g[g].N; K[g] = gp.g[g].K;
    AOD_CHECK_ARG(gp.g[g].x3 == gp.g[0].x3, "wgrad_grouped: members of different precision modes");
  }
  const int x3 = gp.g[0].x3;
  const int big = wgrad_plan_group(n, M, N, K, tn, tk, sp, rps, x3);
  AOD_CHECK_ARG(big >= 0, "wgrad_grouped: the members take different tile forms (aod_conv2d_wgrad_group_plan tells)");
  int wg = 0;
  for (int g = 0; g < n; ++g) {
    WgradParams& p = gp.g[g];
    AOD_CHECK_ARG(sp[g] <= nslabs[g], "wgrad_grouped: member %d needs %d slabs, %d provided", g, sp[g], nslabs[g]);
    AOD_CHECK_ARG(slab_stride[g] >= (long long)p.N * p.K / (p.x3 ? 4 : 1), "wgrad_grouped: slab stride smaller than N*K (x3: N/2 * K/2)");
    p.tiles_n = tn[g]; p.tiles_k = tk[g]; p.splits = sp[g]; p.rows_per_split = rps[g]; p.slab_stride = slab_stride[g];
    gp.wg0[g] = wg;
    wg += tn[g] * tk[g] * sp[g];
  }
  for (int g = n; g <= WG_MAXG; ++g) gp.wg0[g] = g == n ? wg : 0x7fffffff;
  gp.n = n;
  wgrad_attrs();
  if (x3 && big == 2) hipLaunchKernelGGL((conv_wgrad_grouped_x3w_kernel<8, 4>), dim3(wg), dim3(512), 131072 + 4096, (hipStream_t)stream, gp);
  else if (x3 && big == 3) hipLaunchKernelGGL((conv_wgrad_grouped_x3w_kernel<4, 2>), dim3(wg), dim3(256), 65536 + 4096, (hipStream_t)stream, gp);
  else if (x3 && big) hipLaunchKernelGGL((conv_wgrad_grouped_kernel<8, 2, 2, true>), dim3(wg), dim3(512), 131072 + 4096, (hipStream_t)stream, gp);
  else if (x3) hipLaunchKernelGGL((conv_wgrad_grouped_kernel<4, 1, 1, true>), dim3(wg), dim3(256), 65536 + 4096, (hipStream_t)stream, gp);
  else if (big) hipLaunchKernelGGL((conv_wgrad_grouped_kernel<8, 2, 2>), dim3(wg), dim3(512), 131072 + 4096, (hipStream_t)stream, gp);
  else hipLaunchKernelGGL((conv_wgrad_grouped_kernel<8, 1, 1>), dim3(wg), dim3(512), 65536 + 4096, (hipStream_t)stream, gp);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_conv2d_wgrad(const aod_conv_desc_t* d, const void* x, const void* dz, float* dw, const void* row_table,
                                aod_stream_t stream) {
  return wgrad_launch(d, x, dz, dw, 0, 0, row_table, stream);
}

extern "C" int aod_conv2d_wgrad_splits(const aod_conv_desc_t* d) {
  if (!d || d->transposed) return 0;
  ConvKParams cp;
  memset(&cp, 0, sizeof(cp));
  if (fill_params(d, cp) || cp.M == 0) return 0;
  int tn, tk, splits, rps, big;
  wgrad_plan(cp.M, cp.N, cp.K, true, tn, tk, splits, rps, big, 49152, cp.x3);
  return splits;
}

extern "C" int aod_conv2d_wgrad_slabs(const aod_conv_desc_t* d, const void* x, const void* dz, float* slabs, int nslabs, int64_t slab_stride,
                                      const void* row_table, aod_stream_t stream) {
  AOD_CHECK_ARG(nslabs >= 1 && slab_stride > 0, "wgrad_slabs: nslabs / slab_stride");
  return wgrad_launch(d, x, dz, slabs, slab_stride, nslabs, row_table, stream);
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
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, bf16_t* __restrict__ o, int O, int I, int RS,
                                    int Opad) {
  const long long n = (long long)Opad * RS * I;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int oo = i % Opad; const long long r1 = i / Opad; const int rs = r1 % RS; const int c = r1 / RS;
    o[i] = (oo < O) ? (bf16_t)(w[((long long)oo * I + c) * RS + rs] * (scale ? scale[oo] : 1.f)) : (bf16_t)0.f;
  }
}
// One block per output channel o: g[o][c][rs] (+)= scale[o] * dw[o][rs][c], and wdot[o] = <w[o], dw[o]>.
// With y = (z - mean) * invstd * gamma + beta and z = <w[o], patch>, the BN-scale gradient needs sum_m gm[m,o] * z[m,o]
// = <w[o], sum_m gm[m,o] * patch(m)> = <w[o], dW_gm[o]>: the pre-BN activations never have to be stored or re-read.
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(float* __restrict__ dw, float* __restrict__ g, const float* __restrict__ scale,
                                                          const float* __restrict__ w, float* __restrict__ wdot, const float* __restrict__ bn_s1,
                                                          const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd, int O, int I,
                                                          int RS, int Ipad, int accumulate, int clear) {
  __shared__ float red[4];
  const int oo = blockIdx.x;
  const int n = I * RS;
  const float sc = scale ? scale[oo] : 1.f;
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int rs = i % RS, c = i / RS;
    const long long si = ((long long)oo * RS + rs) * Ipad + c;
    const long long gi = (long long)oo * n + i;
    const float v = dw[si];
    if (clear) dw[si] = 0.f;          // hand the accumulator back all-zero: no separate memset launch per conv
    if (w) dot += w[gi] * v;
    g[gi] = accumulate ? g[gi] + v * sc : v * sc;
  }
  if (wdot) {
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float d = red[0] + red[1] + red[2] + red[3];
      wdot[oo] = bn_s1 ? bn_invstd[oo] * (d - bn_mean[oo] * bn_s1[oo]) : d;      // BN weight gradient when the statistics are given
    }
  }
}
// Slab form: dw = [nslabs][Opad][RS][Ipad] partial sums written by conv_wgrad_kernel's splits; one block per output channel adds the
// slabs IN ORDER (deterministic), reading each slab row with consecutive lanes on consecutive channels, transposes [rs][c] -> [c][rs]
// through LDS and writes the OIHW gradient with consecutive lanes on consecutive elements.
constexpr int UNP_CH = 1024;      // channels per LDS chunk at RS <= 9 taps (9 216 tile entries; fewer channels per chunk for 10 .. 16 taps)
constexpr int UNP_T = 1024;       // threads: 16 waves per block keep enough slab loads in flight (one block per output channel)
struct UnpackArgs {
  const float* dw; float* g; const float* scale; const float* w; float* wdot; const float* bn_s1; const float* bn_mean; const float* bn_invstd;
  long long slab_stride;
  int nslabs, O, I, RS, Ipad, accumulate;
  int x3;      // (unused: x3 wgrad launches write LOGICAL slabs -- the three bands of an entry are added in the wgrad epilogue)
};
__device__ __forceinline__ void unpack_row(const UnpackArgs& a, int oo, float* tile, float* red) {
  const float* __restrict__ dw = a.dw;
  float* __restrict__ g = a.g;
  const float* __restrict__ w = a.w;
  const int nslabs = a.nslabs, I = a.I, RS = a.RS, Ipad = a.Ipad, accumulate = a.accumulate;
  const long long slab_stride = a.slab_stride;
  const float sc = a.scale ? a.scale[oo] : 1.f;
  float dot = 0.f;
  const int CH = RS <= 9 ? UNP_CH : (UNP_CH * 9) / RS;            // (filters of 10 .. 16 taps -- SSD512's 4 x 4 extra conv -- take narrower chunks of the same tile)
  for (int c0 = 0; c0 < I; c0 += CH) {
    const int nc = min(CH, I - c0);
    for (int i = threadIdx.x; i < nc * RS; i += UNP_T) {
      const int rs = i / nc, c = i - rs * nc;                       // consecutive lanes on consecutive channels of one slab row
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;                 // fixed association: ((s0+s4+..)+(s1+s5+..))+((s2+..)+(s3+..))
      {
      const float* src = dw + ((long long)oo * RS + rs) * Ipad + c0 + c;
      int sl = 0;
      for (; sl + 4 <= nslabs; sl += 4) {
        const float a0 = src[(long long)sl * slab_stride], a1 = src[(long long)(sl + 1) * slab_stride];
        const float a2 = src[(long long)(sl + 2) * slab_stride], a3 = src[(long long)(sl + 3) * slab_stride];
        v0 += a0; v1 += a1; v2 += a2; v3 += a3;
      }
      for (; sl < nslabs; ++sl) v0 += src[(long long)sl * slab_stride];
      }
      tile[c * RS + rs] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    const long long g0 = ((long long)oo * I + c0) * RS;
    for (int i = threadIdx.x; i < nc * RS; i += UNP_T) {
      const float v = tile[i];
      if (w) dot += w[g0 + i] * v;
      g[g0 + i] = accumulate ? g[g0 + i] + v * sc : v * sc;
    }
    __syncthreads();
  }
  if (a.wdot) {
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      float d = 0.f;
      for (int k = 0; k < UNP_T / 64; ++k) d += red[k];
      a.wdot[oo] = a.bn_s1 ? a.bn_invstd[oo] * (d - a.bn_mean[oo] * a.bn_s1[oo]) : d;
    }
  }
}
__global__ __launch_bounds__(UNP_T) void unpack_wgrad_slabs_kernel(const UnpackArgs a) {
  __shared__ float tile[UNP_CH * 9 + 1];
  __shared__ float red[UNP_T / 64];
  unpack_row(a, blockIdx.x, tile, red);
}
// the unpacks of a grouped wgrad launch in one grid: block -> (member, output channel)
struct UnpackGroups { UnpackArgs g[WG_MAXG]; int blk0[WG_MAXG + 1]; };
__global__ __launch_bounds__(UNP_T) void unpack_wgrad_slabs_grouped_kernel(const UnpackGroups gp) {
  __shared__ float tile[UNP_CH * 9 + 1];
  __shared__ float red[UNP_T / 64];
  const int b = blockIdx.x;
  if (b < gp.blk0[1]) unpack_row(gp.g[0], b, tile, red);
  else if (b < gp.blk0[2]) unpack_row(gp.g[1], b - gp.blk0[1], tile, red);
  else if (b < gp.blk0[3]) unpack_row(gp.g[2], b - gp.blk0[2], tile, red);
  else unpack_row(gp.g[3], b - gp.blk0[3], tile, red);
}
// ---- batched parameter preparation: ONE launch re-derives, for every registered conv layer, the folded eval-BN vectors
// (scale = gamma * rsqrt(var + eps), shift = beta - mean * scale, invstd) and both packed bf16 weight images (forward
// [O][R][S][Ipad]; dgrad [I][R][S][Opad] with the BN scale folded in).  After an optimizer step every trainable layer is stale; doing
// this per layer costs ~270 launches of a few microseconds each per iteration.
struct PrepItem {            // 16 x 8 bytes, filled by the host into a device table
  const float* w; const float* gamma; const float* beta; const float* mean; const float* var;
  bf16_t* wf; bf16_t* wd; float* scale; float* shift; float* invstd;
  int O, I, RS, Ipad, Opad, blk0;      // blk0: first block of this item
  float eps; int flags;                // flags bit 0: X3 images (X-layout along the packed channel axis: Ipad / Opad are then the PHYSICAL
  long long pad2_[2];                  // widths 2 * ceil32(I) / 2 * ceil32(O); head = bf16(v), tail = bf16(v - head), v = w (* BN scale, dgrad))
};
// One block = one 32 (out channels) x 32 (in channels) tile of one layer, all R*S taps: the fp32 master weights are read ONCE, coalesced
// (for a fixed output channel the (c, tap) run is contiguous), staged in LDS and written out as both packed images in 64-byte runs.
// Layers with more than 9 taps (the frozen 7x7 stem) take the element-wise path.
constexpr int PREP_T = 32, PREP_RS = 9;
__global__ __launch_bounds__(256) void param_prep_kernel(const PrepItem* __restrict__ items, int nitems) {
  __shared__ float tile[PREP_RS][PREP_T][PREP_T + 1];     // used as a linear [32][32*RS | 1] array
  int lo = 0, hi = nitems - 1;
  while (lo < hi) {      // block -> item: binary search over blk0 (items are in block order)
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PrepItem it = items[lo];
  const int lb = blockIdx.x - it.blk0;
  const int t = threadIdx.x;
  if (it.RS > PREP_RS) {
    const long long nf = (long long)it.O * it.RS * it.Ipad, nd = it.wd ? (long long)it.I * it.RS * it.Opad : 0;
    for (int u = 0; u < 8; ++u) {
      const long long i = (long long)lb * 2048 + u * 256 + t;
      const bool x3 = it.flags & 1;
      if (i < nf) {
        const int cp = i % it.Ipad; const long long r1 = i / it.Ipad; const int rs = r1 % it.RS; const int oo = r1 / it.RS;
        const int c = x3 ? ((cp >> 6) << 5) + (cp & 31) : cp;              // logical channel of physical column cp
        const float v = (c < it.I) ? it.w[((long long)oo * it.I + c) * it.RS + rs] : 0.f;
        const bf16_t h = (bf16_t)v;
        it.wf[i] = (x3 && (cp & 32)) ? (bf16_t)(v - (float)h) : h;
      }
      if (i < nd) {
        const int op = i % it.Opad; const long long r1 = i / it.Opad; const int rs = r1 % it.RS; const int c = r1 / it.RS;
        const int oo = x3 ? ((op >> 6) << 5) + (op & 31) : op;
        float v = 0.f;
        if (oo < it.O) { v = it.w[((long long)oo * it.I + c) * it.RS + rs]; if (it.gamma) v *= it.gamma[oo] * rsqrtf(it.var[oo] + it.eps); }
        const bf16_t h = (bf16_t)v;
        it.wd[i] = (x3 && (op & 32)) ? (bf16_t)(v - (float)h) : h;
      }
      if (it.gamma && i < it.O) {
        const float inv = rsqrtf(it.var[i] + it.eps), sc = it.gamma[i] * inv;
        it.invstd[i] = inv; it.scale[i] = sc; it.shift[i] = it.beta[i] - it.mean[i] * sc;
      }
    }
    return;
  }
  const bool x3 = it.flags & 1;
  const int Il = x3 ? it.Ipad >> 1 : it.Ipad, Ol = x3 ? it.Opad >> 1 : it.Opad;      // logical (padded) channel counts of the two images
  const int tiles_c = (Il + PREP_T - 1) / PREP_T;
  const int to = lb / tiles_c, tc = lb - to * tiles_c;
  const int o0 = to * PREP_T, c0 = tc * PREP_T;
  const int RS = it.RS;
  const int run = PREP_T * RS;                 // floats of one output channel inside the tile: (c0 .. c0+32) x taps, contiguous in OIHW
  const int pitch = run | 1;                   // odd LDS pitch: column reads (dgrad image) stay conflict-free
  float* tl = &tile[0][0][0];                  // linear [32][pitch]
  const int cvalid = it.I - c0 < PREP_T ? it.I - c0 : PREP_T;     // real input channels in this tile (may be <= 0)
  {
    // the whole 32 x run tile as ONE flat index space of 16-B pieces (run = 32 RS floats is a multiple of 4, so a piece never straddles
    // two output channels): exactly RS pieces per thread, ALL requested before the first one is used (four 4-B loads in flight per thread
    // made every block a chain of nine memory round trips: 118 us per training step for 312 MB)
    const int lim = cvalid * RS;
    const float* src0 = it.w + ((long long)o0 * it.I + c0) * RS;
    const long long ostride = (long long)it.I * RS;
    const bool al = ((((size_t)src0) | ((size_t)ostride * 4)) & 15) == 0;
    f32x4 v[PREP_RS];
    int ol[PREP_RS], idx[PREP_RS];
    const bool full = al && cvalid == PREP_T && o0 + PREP_T <= it.O;      // (block-uniform: no per-thread predicate around the loads)
#pragma unroll
    for (int k = 0; k < PREP_RS; ++k) {
      v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (k < RS) {
        const int e = 4 * (t + k * 256);
        ol[k] = e / run; idx[k] = e - ol[k] * run;
        const float* sp = src0 + ol[k] * ostride + idx[k];
        if (full) v[k] = *reinterpret_cast<const f32x4*>(sp);
        else if (o0 + ol[k] < it.O) {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (idx[k] + j < lim) v[k][j] = sp[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PREP_RS; ++k)
      if (k < RS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[ol[k] * pitch + idx[k] + j] = v[k][j];
      }
  }
  __syncthreads();
  // both images leave the tile as 16-B stores (8 consecutive channels per lane; 2-B stores -- one channel per lane -- held the launch at
  // 2.5 TB/s: 126 us per training step for 312 MB)
  // forward image [O][RS][Ipad]: runs of 32 input channels = 4 octets
  for (int u = t; u < PREP_T * RS * 4; u += 256) {
    const int oct = u & 3, q = u >> 2, ol = q / RS, rs = q - ol * RS;
    const int o = o0 + ol, c = c0 + oct * 8;
    if (o < it.O && c < Il) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)tl[ol * pitch + (oct * 8 + j) * RS + rs];
      if (!x3) *reinterpret_cast<bf16x8*>(it.wf + ((long long)o * RS + rs) * it.Ipad + c) = v;
      else {         // this tile's 32 channels = one head band + one tail band of the X-layout
        bf16x8 l;
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = (bf16_t)(tl[ol * pitch + (oct * 8 + j) * RS + rs] - (float)v[j]);
        bf16_t* d = it.wf + ((long long)o * RS + rs) * it.Ipad + 2 * c0 + oct * 8;
        *reinterpret_cast<bf16x8*>(d) = v;
        *reinterpret_cast<bf16x8*>(d + 32) = l;
      }
    }
  }
  // dgrad image [I][RS][Opad] (x BN scale): runs of 32 output channels = 4 octets
  if (it.wd) {
    __shared__ float s_sc[PREP_T];
    if (t < PREP_T) {
      const int o = o0 + t;
      s_sc[t] = (it.gamma && o < it.O) ? it.gamma[o] * rsqrtf(it.var[o] + it.eps) : 1.f;
    }
    __syncthreads();
    for (int u = t; u < PREP_T * RS * 4; u += 256) {
      const int oct = u & 3, q = u >> 2, cl = q / RS, rs = q - cl * RS;
      const int c = c0 + cl, o = o0 + oct * 8;
      if (c < it.I && o < Ol) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16_t)(tl[(oct * 8 + j) * pitch + cl * RS + rs] * s_sc[oct * 8 + j]);
        if (!x3) *reinterpret_cast<bf16x8*>(it.wd + ((long long)c * RS + rs) * it.Opad + o) = v;
        else {
          bf16x8 l;
#pragma unroll
          for (int j = 0; j < 8; ++j) l[j] = (bf16_t)(tl[(oct * 8 + j) * pitch + cl * RS + rs] * s_sc[oct * 8 + j] - (float)v[j]);
          bf16_t* d = it.wd + ((long long)c * RS + rs) * it.Opad + 2 * o0 + oct * 8;
          *reinterpret_cast<bf16x8*>(d) = v;
          *reinterpret_cast<bf16x8*>(d + 32) = l;
        }
      }
    }
  }
  if (it.gamma && tc == 0 && t < PREP_T && o0 + t < it.O) {
    const int o = o0 + t;
    const float inv = rsqrtf(it.var[o] + it.eps), sc = it.gamma[o] * inv;
    it.invstd[o] = inv; it.scale[o] = sc; it.shift[o] = it.beta[o] - it.mean[o] * sc;
  }
}
extern "C" int aod_param_prep(const void* items_dev, int nitems, int total_blocks, aod_stream_t stream) {
  AOD_CHECK_ARG(items_dev && nitems >= 0 && total_blocks >= 0, "param_prep: bad args");
  if (nitems == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(param_prep_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)items_dev, nitems);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_param_prep_item_bytes(void) { return (int)sizeof(PrepItem); }

static inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

extern "C" int aod_pack_weight_fwd(const float* w, void* o, int O, int I, int R, int S, int Ipad, aod_stream_t stream) {
  AOD_CHECK_ARG(w && o && Ipad >= I && Ipad % 8 == 0, "pack_weight_fwd: bad args");
  hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(grid_for((long long)O * R * S * Ipad)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)o, O, I, R * S, Ipad);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_pack_weight_dgrad(const float* w, void* o, int O, int I, int R, int S, int Opad, const float* scale, aod_stream_t stream) {
  AOD_CHECK_ARG(w && o && Opad >= O && Opad % 8 == 0, "pack_weight_dgrad: Opad must be a multiple of 8 and >= O");
  hipLaunchKernelGGL(pack_w_dgrad_kernel, dim3(grid_for((long long)Opad * R * S * I)), dim3(256), 0, (hipStream_t)stream, w, scale, (bf16_t*)o, O, I, R * S, Opad);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_unpack_wgrad_slabs(const float* dw_slabs, int nslabs, int64_t slab_stride, float* g, int O, int I, int R, int S, int Ipad,
                                      int accumulate, const float* scale, const float* w_oihw, float* wdot, const float* bn_s1,
                                      const float* bn_mean, const float* bn_invstd, aod_stream_t stream) {
  AOD_CHECK_ARG(dw_slabs && g && nslabs >= 1 && slab_stride > 0, "unpack_wgrad_slabs: null / empty");
  AOD_CHECK_ARG(R * S <= 16, "unpack_wgrad_slabs: at most 16 taps (larger filters take aod_unpack_wgrad)");
  AOD_CHECK_ARG(!wdot || w_oihw, "unpack_wgrad_slabs: wdot needs the weights");
  AOD_CHECK_ARG(!bn_s1 || (wdot && bn_mean && bn_invstd), "unpack_wgrad_slabs: BN mode needs wdot, mean and invstd");
  if (O == 0) return 0;
  UnpackArgs a;
  a.dw = dw_slabs; a.g = g; a.scale = scale; a.w = w_oihw; a.wdot = wdot; a.bn_s1 = bn_s1; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd;
  a.slab_stride = slab_stride; a.nslabs = nslabs; a.O = O; a.I = I; a.RS = R * S; a.Ipad = Ipad; a.accumulate = accumulate & 1; a.x3 = (accumulate >> 1) & 1;
  hipLaunchKernelGGL(unpack_wgrad_slabs_kernel, dim3(O), dim3(UNP_T), 0, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_unpack_wgrad_slabs_grouped(int n, const float* const* dw_slabs, const int32_t* nslabs, const int64_t* slab_stride,
                                              float* const* g, const int32_t* O, const int32_t* I, const int32_t* R, const int32_t* S,
                                              const int32_t* Ipad, const int32_t* accumulate, const float* const* scale,
                                              const float* const* w_oihw, float* const* wdot, const float* const* bn_s1,
                                              const float* const* bn_mean, const float* const* bn_invstd, aod_stream_t stream) {
  AOD_CHECK_ARG(n >= 1 && n <= WG_MAXG && dw_slabs && nslabs && slab_stride && g && O && I && R && S && Ipad && accumulate && scale &&
                w_oihw && wdot && bn_s1 && bn_mean && bn_invstd, "unpack_wgrad_slabs_grouped: 1..4 members, no null arrays");
  UnpackGroups gp;
  memset(&gp, 0, sizeof(gp));
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    AOD_CHECK_ARG(dw_slabs[i] && g[i] && nslabs[i] >= 1 && slab_stride[i] > 0 && O[i] >= 1, "unpack_wgrad_slabs_grouped: null / empty member");
    AOD_CHECK_ARG(R[i] * S[i] <= 16, "unpack_wgrad_slabs_grouped: at most 16 taps");
    AOD_CHECK_ARG(!wdot[i] || w_oihw[i], "unpack_wgrad_slabs_grouped: wdot needs the weights");
    AOD_CHECK_ARG(!bn_s1[i] || (wdot[i] && bn_mean[i] && bn_invstd[i]), "unpack_wgrad_slabs_grouped: BN mode needs wdot, mean and invstd");
    UnpackArgs& a = gp.g[i];
    a.dw = dw_slabs[i]; a.g = g[i]; a.scale = scale[i]; a.w = w_oihw[i]; a.wdot = wdot[i]; a.bn_s1 = bn_s1[i]; a.bn_mean = bn_mean[i];
    a.bn_invstd = bn_invstd[i];
    a.slab_stride = slab_stride[i]; a.nslabs = nslabs[i]; a.O = O[i]; a.I = I[i]; a.RS = R[i] * S[i]; a.Ipad = Ipad[i]; a.accumulate = accumulate[i] & 1; a.x3 = (accumulate[i] >> 1) & 1;
    gp.blk0[i] = blk;
    blk += O[i];
  }
  for (int i = n; i <= WG_MAXG; ++i) gp.blk0[i] = i == n ? blk : 0x7fffffff;
  hipLaunchKernelGGL(unpack_wgrad_slabs_grouped_kernel, dim3(blk), dim3(UNP_T), 0, (hipStream_t)stream, gp);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_unpack_wgrad(float* dw, float* g, int O, int I, int R, int S, int Ipad, int accumulate, int clear_src, const float* scale,
                                const float* w_oihw, float* wdot, const float* bn_s1, const float* bn_mean, const float* bn_invstd,
                                aod_stream_t stream) {
  AOD_CHECK_ARG(dw && g, "unpack_wgrad: null");
  AOD_CHECK_ARG(!wdot || w_oihw, "unpack_wgrad: wdot needs the weights");
  AOD_CHECK_ARG(!bn_s1 || (wdot && bn_mean && bn_invstd), "unpack_wgrad: BN mode needs wdot, mean and invstd");
  if (O == 0) return 0;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, dw, g, scale, w_oihw, wdot, bn_s1, bn_mean, bn_invstd, O, I,
                     R * S, Ipad, accumulate, clear_src);
  AOD_LAUNCH_CHECK();
  return 0;
}
