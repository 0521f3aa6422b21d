// Whole 64-plane ResNet bottleneck forward in ONE kernel, REFERENCE-PRECISION mode (conv.hip "X3"): the x3 twin of bottleneck.hip for the
// frozen layer 1 (mmdet/models/backbones/resnet.py:262-301 with frozen_stages >= 1: forward-only in the training step and in the scoring pass)
//   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + res)
// on X-layout rows (bf16 head + tail pairs, [h32 | l32] per 32 channels), every product as xh*wh + xl*wh + xh*wl in one fp32 accumulator.
// As three launches the stage is at the HBM roofline and moves an activation of 4 B per element five times per block (x for conv1, x again
// as the residual, y, and the two 64-channel intermediates written and re-read); fused, x and y cross HBM once.
//
// One workgroup (8 waves) = an 8 x 16 pixel tile of one image (the X rows are twice as wide as the bf16 kernel's: half its tile):
//   phase 1  conv1 on the 10 x 18 HALO (180 pixels, 12 blocks of 16): x streamed through LDS in 64-column K-steps (= 32 channels: heads in
//            k-block 0, tails in k-block 1) by LDS-DMA, two stages in flight under counted waits; t1 = relu(bn1(.)) -> LDS as X rows (two
//            sub-images of 32 channels), ZERO outside the image (conv2 pads t1);
//   phase 2  conv2 as 18 (tap, channel group) K-steps: pixel fragments gathered from t1 in LDS, the 144 KB X filter streamed through an
//            8-slot ring in groups of four steps; t2 = relu(bn2(.)) -> LDS over t1;
//   phase 3  conv3 from t2 against the X filter (64 KB, fetched into the ring's slots as they fall free), two halves of 128 output
//            channels; epilogue in registers: + residual (head + tail), ReLU, head / tail stores.
// DS form (the stage's FIRST block, Cin = 64): the residual is the block's downsample branch bn_d(conv_d_1x1(x)) (resnet.py:291-292), computed here
// as well instead of being written (268 MB at 16 x 128 x 128) by a launch of its own and read back: per half of 128 output channels, the
// wave's 16 pixels of x (fragments straight from global memory: L2 hits, the halo tile was just read) against half of the X filter [256][128]
// staged in a 32 KB LDS region of its own; BN, rounding to head + tail as the separate launch would store it -> the residual registers.
// Products are computed transposed (filter fragment = the MFMA's A operand) with the paired-block row permutation of bottleneck.hip, so a
// lane holds 8 consecutive channels of one pixel and every intermediate / residual / output piece is 16 B.
#include "common.h"

namespace {

struct Bnx3Args {
  const bf16_t* x;       // X rows [B*H*W][2 * Cin]
  const bf16_t* w1;      // X filter [64][2 * Cin]
  const bf16_t* w2;      // X filter [64][9][128]
  const bf16_t* w3;      // X filter [256][128]
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  const bf16_t* res;     // X rows [B*H*W][512] (may alias x when Cin == 256); unused in the DS form
  const bf16_t* wd; const float* sd; const float* bd;      // DS form: X filter [256][128] and folded BN of the downsample conv
  bf16_t* y;             // X rows [B*H*W][512]
  int B, H, W, CP, tiles_y, tiles_x;      // CP = 2 * Cin: physical width of x
  int st3;                                // phase 1 on three stages (AOD_B64X3_ST3=0: two)
};

constexpr int TH = 8, TW = 16, HW_ = 18, HPIX = (TH + 2) * HW_;      // 180 halo pixels
constexpr int M1 = 192;                                // padded to 12 row blocks of 16
constexpr int XBUF = M1 * 128;                         // one K-step of the halo tile: 24 KB
constexpr int OFF_W1 = 2 * XBUF;                       // 49152: [2][64 rows][128 B]
constexpr int OFF_T1 = 65536;                          // [2 sub-images][192][128 B] = 48 KB; later t2 [2][128][128 B]
constexpr int T1SUB = M1 * 128, T2SUB = TH * TW * 128;
constexpr int OFF_VEC = OFF_T1 + 2 * T1SUB;            // 114688
constexpr int NVEC = 1280;                             // s1 b1 s2 b2 | s3 | b3 | sd | bd
constexpr int OFF_WD = OFF_VEC + NVEC * 4;             // 119808: DS form, half of the downsample filter [2 channel groups][128 rows][128 B]
constexpr int LDS_BYTES = OFF_WD, LDS_BYTES_DS = OFF_WD + 32768;      // 119808 / 152576
constexpr int SLOT = 8192;                             // conv2 filter ring: 8 slots of [64 rows][128 B] over the phase-1 stages
constexpr int W3SUB = 256 * 128;                       // conv3 filter: 2 sub-images [256][128 B] over the ring
static_assert(OFF_W1 + 2 * 8192 <= OFF_T1 && 8 * SLOT <= OFF_T1 && 2 * W3SUB <= OFF_T1 && 2 * T2SUB <= 2 * T1SUB && LDS_BYTES_DS <= 160 * 1024, "LDS map");

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#ifdef AOD_TILE_TIMING
__device__ unsigned long long* g_b64_stamps = nullptr;      // debug build (tools/dbg/b64x3_timing.py): phase stamps of every tile, 100 MHz wall clock
#define BSTAMP(k) do { if (g_b64_stamps && threadIdx.x == 0) g_b64_stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define BSTAMP(k) do {} while (0)
#endif

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }

// v[8] -> head and tail pieces
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { h[j] = (bf16_t)v[j]; l[j] = (bf16_t)(v[j] - (float)h[j]); }
}

template <bool DS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck64x3_fwd_kernel(const Bnx3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, lane = t & 63;
  BSTAMP(0);
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const long long npix = (long long)p.B * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(npix * p.CP * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 64 * p.CP * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 9 * 128 * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, 256 * 128 * 2, 0x00020000);
  const auto rsrc_res = __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? p.x : p.res), 0, DS ? (int)(npix * p.CP * 2) : (int)(npix * 1024), 0x00020000);
  const auto rsrc_wd = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, DS ? 256 * 128 * 2 : 0, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(npix * 1024), 0x00020000);
  constexpr unsigned OOB = 0xf0000000u;
  // LDS-DMA lane roles (bottleneck.hip): one wave-instruction fills 8 rows x 8 chunks; lane -> row (lane >> 3) of the group, slot lane & 7
  const int drow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int kcw = kc ^ (((uw >> 1) & 1) << 1);          // filter images (wswz): rows 8 (uw + 8 i) + drow have (row >> 4) & 1 = (uw >> 1) & 1

  auto halo_pix = [&](int h, int& y, int& x) -> bool {
    const int hy = h / HW_, hx = h - hy * HW_;
    y = ty0 - 1 + hy; x = tx0 - 1 + hx;
    return h < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
  };
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  {
    float v;
    if (t < 64) v = p.s1[t];
    else if (t < 128) v = p.b1[t - 64];
    else if (t < 192) v = p.s2[t - 128];
    else if (t < 256) v = p.b2[t - 192];
    else v = p.s3[t - 256];
    vec[t] = v;
    if (t < 256) vec[512 + t] = p.b3[t];
    if constexpr (DS) {
      if (t < 256) vec[768 + t] = p.sd[t]; else vec[768 + t] = p.bd[t - 256];
    }
  }
  const float* const vs1 = vec, * const vb1 = vec + 64, * const vs2 = vec + 128, * const vb2 = vec + 192, * const vs3 = vec + 256, * const vb3 = vec + 512;
  const float* const vsd = vec + 768, * const vbd = vec + 1024;
  // DS: half h of the downsample filter, rows 128 h .. + 127 as two channel-group images [128][128 B] (4 instructions per wave).  Half 0 goes
  // out FIRST: it is the oldest vector-memory operation of the wave, so every counted wait below (which names the YOUNGER operations that
  // may stay in flight) covers it
  const unsigned wdlane = (unsigned)((8 * uw + drow) * 256 + kcw * 16);
  auto issueD = [&](int half) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wd, (__attribute__((address_space(3))) void*)(smem + OFF_WD + g * 16384 + (uw + 8 * i) * 1024), 16,
                                                 wdlane + (unsigned)((half * 128 + i * 64) * 256) + (unsigned)g * 128u, 0, 0, 0);
  };
  if constexpr (DS) issueD(0);

  // ------------------------------------------------------------------ phase 1: t1 = relu(bn1(conv1(x))) on the halo
  unsigned xoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int h = 8 * (uw + 8 * i) + drow;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * p.CP + kc * 8) * 2) : OOB;
  }
  unsigned w1off = (unsigned)(((8 * uw + drow) * p.CP + kcw * 8) * 2);
  const int nk1 = p.CP >> 6;
  // THREE stages in flight (a K-step is 36 MFMAs per SIMD, a quarter of a memory round trip: with two stages every step waited for its
  // loads): stages 0 and 1 where they always were, stage 2 in the t1 region, which is written only after the last K-step
  auto xstage = [&](int buf) -> char* { return smem + (buf < 2 ? buf * XBUF : OFF_T1); };
  auto wstage = [&](int buf) -> char* { return smem + (buf < 2 ? OFF_W1 + buf * 8192 : OFF_T1 + XBUF); };
  static_assert(OFF_T1 + XBUF + 8192 <= OFF_VEC, "third phase-1 stage inside the t1 region");
  auto issue1 = [&](int buf) {          // 3 x-halo instructions + 1 filter instruction per wave and stage
    char* xs = xstage(buf);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned off = xoff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(xs + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      xoff[i] += 128;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(wstage(buf) + uw * 1024), 16, w1off, 0, 0, 0);
    w1off += 128;
  };
  const bool two = uw < 4;                // 12 pixel blocks over 8 waves: waves 0-3 own blocks uw and uw + 8, waves 4-7 block uw
  f32x4 acc1[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nst = p.st3 ? 3 : 2;
  issue1(0);
  if (nk1 > 1) issue1(1);
  if (nst > 2 && nk1 > 2) issue1(2);
  int buf = 0;
  for (int kt = 0; kt < nk1; ++kt) {
    // younger stages of this wave that may stay in flight (4 instructions each)
    const int young = min(nk1 - 1 - kt, nst - 1);
    if (young >= 2) wait_vm<8>(); else if (young == 1) wait_vm<4>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage kt has landed
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = xstage(buf);
    const char* ws = wstage(buf);
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), lq));
      wl[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), 4 + lq));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 0 || two) {
        const int row = (uw + 8 * i) * 16 + lr;
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xs + swz(row, lq)), xl = *reinterpret_cast<const bf16x8*>(xs + swz(row, 4 + lq));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (the order of conv_igemm_kernel<X3>: heads x heads, activation tails x filter heads, activation heads x filter tails)
          acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh, acc1[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xl, acc1[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh, acc1[i][j], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading stage kt: its buffer can be refilled
    if (kt + nst < nk1) issue1(buf);
    buf = buf == nst - 1 ? 0 : buf + 1;
  }
  BSTAMP(1);
  // the stages are dead: conv2's first eight filter steps stream into the ring under epilogue 1 (step s = tap * 2 + channel group)
  const unsigned w2lane = (unsigned)(((8 * uw + drow) * (9 * 128) + kcw * 8) * 2);
  auto issue2 = [&](int s) {              // one instruction per wave: rows 8 uw .. + 7 of the [64][128 B] slice of step s
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(smem + (s & 7) * SLOT + uw * 1024), 16,
                                             w2lane + (unsigned)s * 128u, 0, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < 8; ++s) issue2(s);
  {
    char* t1 = smem + OFF_T1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 0 || two) {
        const int h = (uw + 8 * i) * 16 + lr;
        int y, x;
        const bool ok = halo_pix(h, y, x);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int c = jp * 32 + lq * 8;
          float v[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f;
          }
          bf16x8 oh, ol;
          split8(v, oh, ol);
          *reinterpret_cast<bf16x8*>(t1 + jp * T1SUB + swz(h, lq)) = oh;
          *reinterpret_cast<bf16x8*>(t1 + jp * T1SUB + swz(h, 4 + lq)) = ol;
        }
      }
    }
  }

  BSTAMP(2);
  // ------------------------------------------------------------------ phase 2: t2 = relu(bn2(conv2(t1))); wave uw = output row uw of the tile
  f32x4 acc2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned w3lane = (unsigned)((8 * uw + drow) * 256 + kcw * 16);      // byte offset of the lane's chunk in row 8 uw + drow of the [256][128] X filter
  auto issue3 = [&](int g) {              // conv3 filter, channel group g: [256 rows][128 B] into ring slots 4 g .. 4 g + 3 (4 instructions per wave)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(smem + g * W3SUB + (uw + 8 * i) * 1024), 16,
                                               w3lane + (unsigned)(i * 64 * 256) + (unsigned)g * 128u, 0, 0, 0);
  };
  // groups of four steps: group G lives in slots 4 (G & 1) .. + 3; while it is consumed, group G + 1 is in flight
#pragma unroll
  for (int G = 0; G < 5; ++G) {
    const int nst = G < 4 ? 4 : 2;
    // younger vector-memory instructions of this wave that may stay in flight: the next group's
    if (G == 0) wait_vm<4>();
    else if (G == 1) wait_vm<4>();
    else if (G == 2) wait_vm<4>();
    else if (G == 3) wait_vm<2>();
    else wait_vm<4>();                      // (G == 4: the first half of the conv3 filter is younger)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (G == 0: this wave's t1 pieces are written)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < nst; ++k) {
      const int s = 4 * G + k, tap = s >> 1, g = s & 1;
      const int r = tap / 3, q = tap - r * 3;
      const char* ws = smem + (s & 7) * SLOT;
      const char* t1 = smem + OFF_T1 + g * T1SUB;
      const int row = (uw + r) * HW_ + lr + q;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(t1 + swz(row, lq)), al = *reinterpret_cast<const bf16x8*>(t1 + swz(row, 4 + lq));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), lq)), fl = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), 4 + lq));
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ah, acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, al, acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, ah, acc2[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // group G's slots are free
    if (G < 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (4 * (G + 2) + k < 18) issue2(4 * (G + 2) + k);
    } else if (G == 3) issue3(1);                 // slots 4-7 (group 3 consumed): second channel group of the conv3 filter
    else issue3(0);                               // slots 0-3
  }
  BSTAMP(3);
  // (issue order per wave: g2 x4 | g3 x4 | g4 x2 | w3[1] x4 | w3[0] x4 behind the prologue's g0 x4, g1 x4 -- the waits above count the
  // instructions younger than the group being consumed)
  {
    char* t2 = smem + OFF_T1;                     // t1 is dead behind the last barrier
    const int o_ = uw * 16 + lr;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int c = jp * 32 + lq * 8;
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * q + r] = fmaxf(acc2[2 * jp + q][r] * sc[r] + sh[r], 0.f);
      }
      bf16x8 oh, ol;
      split8(v, oh, ol);
      *reinterpret_cast<bf16x8*>(t2 + jp * T2SUB + swz(o_, lq)) = oh;
      *reinterpret_cast<bf16x8*>(t2 + jp * T2SUB + swz(o_, 4 + lq)) = ol;
    }
  }
  // residual pieces of the first half (x was just read by this XCD: L2 hits); 16-B heads and tails of the lane's 8 channels
  const int oy = ty0 + uw, ox = tx0 + lr;
  const unsigned prow = (oy < p.H && ox < p.W) ? (unsigned)((img0 + (long long)oy * p.W + ox) * 1024) : OOB;
  u32x4_t rh[4], rl[4];
  auto load_res = [&](int half) {
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int c = half * 128 + jp * 32 + lq * 8;              // logical channel: X column ((c >> 5) << 6) + (c & 31)
      const unsigned col = (unsigned)((((c >> 5) << 6) + (c & 31)) * 2);
      rh[jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, (int)(prow + col), 0, 0);
      rl[jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, (int)(prow + col + 64u), 0, 0);
    }
  };
  // DS: the wave's 16 pixels of x as MFMA fragments (lane = pixel lr, k-chunk lq): heads and tails of the two 32-channel groups
  u32x4_t xf[4];
  auto load_xf = [&]() {
    const unsigned xrow = (oy < p.H && ox < p.W) ? (unsigned)((img0 + (long long)oy * p.W + ox) * 256) : OOB;
#pragma unroll
    for (int q = 0; q < 4; ++q) xf[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, (int)(xrow + (unsigned)(q * 64 + lq * 16)), 0, 0);
  };
  // ... and the residual of one half from them: rh / rl = head / tail of bn_d(conv_d(x)), the bits the separate launch would have stored
  auto compute_res = [&](int half) {
    f32x4 accd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) accd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const char* ws = smem + OFF_WD + g * 16384;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, xf[2 * g]), al = __builtin_bit_cast(bf16x8, xf[2 * g + 1]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int fr = wrow(j, lr);
        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(ws + wswz(fr, lq)), fl = *reinterpret_cast<const bf16x8*>(ws + wswz(fr, 4 + lq));
        accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ah, accd[j], 0, 0, 0);
        accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, al, accd[j], 0, 0, 0);
        accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, ah, accd[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int c = half * 128 + jp * 32 + lq * 8;
      float v[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vsd + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vbd + c + 4 * h2);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * h2 + r] = accd[2 * jp + h2][r] * sc[r] + sh[r];
      }
      bf16x8 oh, ol;
      split8(v, oh, ol);
      rh[jp] = __builtin_bit_cast(u32x4_t, oh); rl[jp] = __builtin_bit_cast(u32x4_t, ol);
    }
  };
  wait_vm<0>();                                   // the conv3 filter has landed (this wave's part; DS: and half 0 of the downsample filter)
  if constexpr (DS) load_xf(); else load_res(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // t2 and the conv3 filter complete
  __builtin_amdgcn_sched_barrier(0);
  BSTAMP(4);

  // ------------------------------------------------------------------ phase 3: y = relu(bn3(conv3(t2)) + res), 2 x 128 output channels
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if constexpr (DS) {
      if (half == 1) {
        // per wave, in issue order: issueD(1) x4 | half 0's stores x8 | load_xf x4 -- the filter half has landed when <= 12 younger ones are out
        wait_vm<12>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      compute_res(half);
      if (half == 0) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // every wave is done reading half 0 of the downsample filter
        issueD(1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x4 acc3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc3[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const char* t2 = smem + OFF_T1 + g * T2SUB;
      const char* ws = smem + g * W3SUB;
      const int row = uw * 16 + lr;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(t2 + swz(row, lq)), al = *reinterpret_cast<const bf16x8*>(t2 + swz(row, 4 + lq));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int fr = half * 128 + wrow(j, lr);
        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(ws + wswz(fr, lq)), fl = *reinterpret_cast<const bf16x8*>(ws + wswz(fr, 4 + lq));
        acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ah, acc3[j], 0, 0, 0);
        acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, al, acc3[j], 0, 0, 0);
        acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, ah, acc3[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int c = half * 128 + jp * 32 + lq * 8;
      const u32x4_t qh = rh[jp], ql = rl[jp];
      float v[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * h2 + r] = acc3[2 * jp + h2][r] * sc[r] + sh[r];
        v[4 * h2 + 0] += __uint_as_float(qh[2 * h2] << 16) + __uint_as_float(ql[2 * h2] << 16);
        v[4 * h2 + 1] += __uint_as_float(qh[2 * h2] & 0xffff0000u) + __uint_as_float(ql[2 * h2] & 0xffff0000u);
        v[4 * h2 + 2] += __uint_as_float(qh[2 * h2 + 1] << 16) + __uint_as_float(ql[2 * h2 + 1] << 16);
        v[4 * h2 + 3] += __uint_as_float(qh[2 * h2 + 1] & 0xffff0000u) + __uint_as_float(ql[2 * h2 + 1] & 0xffff0000u);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
      bf16x8 oh, ol;
      split8(v, oh, ol);
      const unsigned col = (unsigned)((((c >> 5) << 6) + (c & 31)) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), rsrc_y, (int)(prow + col), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), rsrc_y, (int)(prow + col + 64u), 0, 0);
    }
    if (half == 0) { if constexpr (DS) { __builtin_amdgcn_sched_barrier(0); load_xf(); } else load_res(1); }
    BSTAMP(5 + half);
  }
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BSTAMP(7);
#endif
}

}  // namespace

#ifdef AOD_TILE_TIMING
extern "C" int aod_dbg_set_b64_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_b64_stamps), &buf, sizeof(buf)); }
#endif

static int launch_bnx3(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                       const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* res, const void* wd,
                       const float* sd, const float* bd, void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck64x3: null pointer");
  AOD_CHECK_ARG(Cin >= 32 && Cin % 32 == 0 && B >= 1 && H >= 1 && W >= 1, "bottleneck64x3: Cin %d must be a multiple of 32", Cin);
  AOD_CHECK_ARG((long long)B * H * W * (Cin > 256 ? Cin : 256) * 4 < 0xe0000000ll, "bottleneck64x3: operand larger than 3.5 GiB");
  Bnx3Args a;
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.res = (const bf16_t*)res; a.y = (bf16_t*)y;
  a.wd = (const bf16_t*)wd; a.sd = sd; a.bd = bd;
  a.B = B; a.H = H; a.W = W; a.CP = 2 * Cin;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  { const char* e = getenv("AOD_B64X3_ST3"); a.st3 = (e && e[0] == '0') ? 0 : 1; }      // (read per call: tests switch it in-process)
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64x3_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64x3_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_DS);
  }
  if (wd) hipLaunchKernelGGL(bottleneck64x3_fwd_kernel<true>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES_DS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bottleneck64x3_fwd_kernel<false>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_bottleneck64x3_fwd(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                      const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* res,
                                      void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(res, "bottleneck64x3: null pointer");
  return launch_bnx3(x, Cin, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, res, nullptr, nullptr, nullptr, y, stream);
}

extern "C" int aod_bottleneck64x3_ds_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                         const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd,
                                         const float* sd, const float* bd, void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(wd && sd && bd, "bottleneck64x3_ds: null pointer");
  return launch_bnx3(x, 64, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, nullptr, wd, sd, bd, y, stream);
}
