// Whole ResNet bottleneck forward in ONE kernel (inference form: eval-mode BN folded into scale / shift, no tensor kept for a backward):
//   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + res)        (mmdet/models/backbones/resnet.py:262-301)
// for the 64-channel stage (layer1: 16 x 128 x 128 pixels at the bench size; frozen, so forward-only in the training step too).  As three
// launches that stage moves 536 MB per block -- x is read twice (conv1 input, residual) and the two 64-channel intermediates are
// written and re-read -- at the HBM rate; fused, x and y cross HBM once (268 MB).
//
// One workgroup (8 waves) = one 16 x 16 pixel tile of one image:
//   phase 1  conv1 on the 18 x 18 HALO of the tile (conv2 needs t1 one pixel around it; 27 % recompute): x streamed through LDS in
//            64-channel K-steps by LDS-DMA (double buffered), t1 = relu(bn1(.)) -> LDS as bf16, ZERO outside the image (conv2 pads t1);
//   phase 2  conv2 as 9 tap K-steps whose A fragments are gathered from t1 in LDS (row (oy + r) * 18 + ox + s), filter taps streamed
//            by LDS-DMA; t2 = relu(bn2(.)) -> LDS;
//   phase 3  conv3 from t2 in LDS against the whole 256 x 64 filter (fetched into LDS during phase 2), two halves of 128 output channels;
//            epilogue in registers: + residual (8-B loads), ReLU, 8-B stores.
// All three products are computed TRANSPOSED (filter fragment as the MFMA's A operand), so a lane holds 4 consecutive channels of one
// pixel: the intermediates go to LDS with one ds_write_b64 per 16 x 16 block and the output leaves in 8-B pieces (see pointwise.hip).
// DS form (the stage's FIRST block, Cin = 64): the residual is the block's downsample branch bn_d(conv_d_1x1(x)) (resnet.py:291-292), computed in the
// same launch -- the wave's 2 x 16 pixels of x as fragments straight from global memory (L2 hits) against the [256][64] filter, fetched into the
// dead t1 region once conv2 is done; rounded to bf16 as the separate launch would store it -> the residual registers: identical bits, one launch
// and 134 MB of residual traffic less (the twin of bottleneck_x3.hip's DS form).
// LDS images use the 128-B-row XOR swizzle of conv.hip (slot s of row r holds 16-B chunk s ^ ((r >> 1) & 7)).
#include "common.h"

namespace {

struct BnkArgs {
  const bf16_t* x;       // [B*H*W][Cin]
  const bf16_t* w1;      // [64][Cin]
  const bf16_t* w2;      // [64][9][64]  (packed forward form [O][R][S][C])
  const bf16_t* w3;      // [256][64]
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  const bf16_t* res;     // [B*H*W][256] (may alias x when Cin == 256); unused in the DS form
  const bf16_t* wd; const float* sd; const float* bd;      // DS form: packed [256][64] filter and folded BN of the downsample conv
  bf16_t* y;             // [B*H*W][256]
  int B, H, W, Cin, tiles_y, tiles_x;
};

constexpr int TH = 16, TW = 16, HW_ = 18;            // tile and halo side
constexpr int M1 = 336;                               // halo pixels 324, padded to 21 row blocks of 16
constexpr int XBUF = M1 * 128;                        // one 64-channel K-step of the halo tile
constexpr int OFF_W1 = 2 * XBUF;                      // [2][64 rows][128 B]  (86016 .. 102400)
constexpr int OFF_T1 = 106496;                        // [336][128 B]
constexpr int OFF_VEC = OFF_T1 + XBUF;                // fp32 s1 b1 s2 b2 [64] s3 b3 [256] (sd bd [256] in the DS form): 1280 floats
constexpr int LDS_BYTES = OFF_VEC + 1280 * 4;         // 154624
constexpr int OFF_WD = OFF_T1;                        // DS form: [256][128 B] downsample filter over t1 once conv2 has consumed it
constexpr int OFF_W2 = 0;                             // phases 2 / 3 reuse the x / w1 buffers: [9 taps][64][128 B] conv2 filter,
constexpr int OFF_W3 = 9 * 8192;                      // [256][128 B] conv3 filter (73728 .. 106496),
constexpr int OFF_T2 = 0;                             // [256][128 B] t2 over the conv2 filter once its last tap has been consumed
static_assert(OFF_W1 + 2 * 8192 <= OFF_T1 && OFF_W3 + 32768 <= OFF_T1 && OFF_WD + 32768 <= OFF_VEC && LDS_BYTES <= 160 * 1024, "LDS map");

#ifdef AOD_TILE_TIMING
// debug build only (tools/dbg/bn_timing.py): per-workgroup wall-clock stamps (100 MHz) at the phase boundaries
__device__ unsigned long long* g_bn_stamps = nullptr;
#define BSTAMP(k) do { if (g_bn_stamps && threadIdx.x == 0) g_bn_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define BSTAMP(k) do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// filter images: the paired-block row permutation (wrow) makes the 16 lanes of a fragment read rows {0-3, 8-11, 16-19, 24-27} (+4): rows r and
// r + 16 share (r >> 1) & 7 and lie 2048 B apart -- the same banks.  One more row bit in the XOR key separates them (PMC before: 35 % of
// the LDS cycles of the kernel were bank conflicts).
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
// Which output channel sits in MFMA row `lr` of 16-row block j.  The blocks are PAIRED: rows 4q .. 4q + 3 of block 2p are channels
// 32p + 8q + 0..3 and of block 2p + 1 channels 32p + 8q + 4..7, so that in the transposed accumulators lane group q holds the 8
// CONSECUTIVE channels 32p + 8q .. + 7 across the pair: intermediates, residual and output move in 16-B pieces.
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }

template <bool DS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck64_fwd_kernel(const BnkArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, lane = t & 63;
  BSTAMP(0);
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.B * p.H * p.W * p.Cin * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 64 * p.Cin * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 576 * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, 256 * 64 * 2, 0x00020000);
  constexpr unsigned OOB = 0xf0000000u;
  // LDS-DMA lane roles: one wave-instruction fills 8 rows x 8 chunks; lane -> row (lane >> 3) of the group, slot lane & 7, source chunk
  // slot ^ ((row >> 1) & 7) -- the row groups a wave serves are 8 apart, so that chunk is the same for all of them
  const int drow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int kcw = kc ^ (((uw >> 1) & 1) << 1);          // filter images: wswz key (row >> 4) & 1 = (uw >> 1) & 1 for rows 8 (uw + 8 i) + drow

  // validity of halo pixel h (0 .. 323) of this tile: inside the image
  auto halo_pix = [&](int h, int& y, int& x) -> bool {
    const int hy = h / HW_, hx = h - hy * HW_;
    y = ty0 - 1 + hy; x = tx0 - 1 + hx;
    return h < HW_ * HW_ && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
  };

  // folded BN vectors -> LDS once (an ordinary global load beside LDS-DMA makes hipcc drain the DMA queue where the value is used)
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
  const auto rsrc_wd = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, DS ? 256 * 64 * 2 : 0, 0x00020000);

  // ------------------------------------------------------------------ phase 1: t1 = relu(bn1(conv1(x))) on the halo
  unsigned xoff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int h = 8 * (uw + 8 * i) + drow;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * p.Cin + kc * 8) * 2) : OOB;
  }
  unsigned w1off = (unsigned)(((8 * uw + drow) * p.Cin + kcw * 8) * 2);
  const int nk1 = p.Cin >> 6;
  auto issue1 = [&](int buf) {
    char* xs = smem + buf * XBUF;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (uw + 8 * i < M1 / 8) {
        const unsigned off = xoff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(xs + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
        xoff[i] += 128;
      }
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(smem + OFF_W1 + buf * 8192 + uw * 1024), 16, w1off, 0, 0, 0);
    w1off += 128;
  };
  f32x4 acc1[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // two K-steps in flight: a K-step's MFMAs are ~0.2 us, a load ~2 us -- the counted wait leaves the next stage's LDS-DMA outstanding
  // across the barrier (a __syncthreads() would drain it); per wave and stage: 5 or 6 x instructions + 1 filter instruction
  issue1(0);
  if (nk1 > 1) issue1(1);
  BSTAMP(1);
  for (int kt = 0; kt < nk1; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk1) { if (uw < 2) wait_vm<7>(); else wait_vm<6>(); }
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage kt has landed
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = smem + buf * XBUF;
    const char* ws = smem + OFF_W1 + buf * 8192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (uw + 8 * i < M1 / 16) xf[i] = *reinterpret_cast<const bf16x8*>(xs + swz((uw + 8 * i) * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (uw + 8 * i < M1 / 16) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading stage kt: its buffer can be refilled
    if (kt + 2 < nk1) issue1(buf);
  }
  BSTAMP(2);
  // the x / w1 buffers are dead: the whole conv2 filter (9 taps) and the conv3 filter stream in under the epilogue, and the first half of
  // the residual tile is requested (x was just read by this CU's XCD: L2 hits)
  {
    const unsigned w2off = (unsigned)(((8 * uw + drow) * 576 + kcw * 8) * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(smem + OFF_W2 + tap * 8192 + uw * 1024), 16,
                                               w2off + (unsigned)tap * 128u, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (unsigned)(((8 * (uw + 8 * i) + drow) * 64 + kcw * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(smem + OFF_W3 + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
    }
  }
  // (branch-free: pixels outside the image get an out-of-range buffer offset -- a conditional load makes hipcc wait for each value
  // where the two paths merge, 32 serial L2 round trips per tile)
  const unsigned y_bytes = (unsigned)((long long)p.B * p.H * p.W * 256 * 2);
  const auto rsrc_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)y_bytes, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)y_bytes, 0x00020000);
  unsigned prow[2];         // byte offset of the lane's pixel row in [B*H*W][256] bf16
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int y = ty0 + 2 * uw + i, x = tx0 + lr;
    prow[i] = (y < p.H && x < p.W) ? (unsigned)((img0 + (long long)y * p.W + x) * 512) : OOB;
  }
  // the residual tile (16-B pieces: 2 halves x 2 pixel rows x 4 channel groups per lane) is requested a few pieces per conv2 tap, so
  // that the address processing of these row-strided loads runs under the MFMAs instead of stalling the wave in one burst
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t rv[2][2][4];
  BSTAMP(8);
  {
    char* t1 = smem + OFF_T1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (uw + 8 * i < M1 / 16) {
        const int h = (uw + 8 * i) * 16 + lr;
        int y, x;
        const bool ok = halo_pix(h, y, x);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int c = jp * 32 + lq * 8;
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f);
          }
          *reinterpret_cast<bf16x8*>(t1 + swz(h, jp * 4 + lq)) = o;
        }
      }
    }
  }
  BSTAMP(9);
  __syncthreads();

  BSTAMP(3);
  // ------------------------------------------------------------------ phase 2: t2 = relu(bn2(conv2(t1))), taps gathered from LDS
  f32x4 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s = tap - r * 3;
    if (!DS && tap < 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {       // piece 4 * tap + k -> (half, i, jp)
        const int half = tap >> 1, i = tap & 1, jp = k;
        rv[half][i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, (int)(prow[i] + (unsigned)((half * 128 + jp * 32 + lq * 8) * 2)), 0, 0);
      }
    }
    const char* t1 = smem + OFF_T1;
    const char* ws = smem + OFF_W2 + tap * 8192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], af[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t1 + swz((2 * uw + i + r) * HW_ + lr + s, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc2[i][j], 0, 0, 0);
    }
  }
  __syncthreads();      // t2 overwrites the conv2 filter
  BSTAMP(4);
  // DS: t1 is dead as well -- the downsample filter takes its place (it lands under the t2 epilogue; the barrier behind it drains the queue),
  // and the wave's 2 x 16 pixels of x are requested as fragments (lane = pixel lr, k-chunk lq)
  u32x4_t xfr[2][2];
  if constexpr (DS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (unsigned)(((8 * (uw + 8 * i) + drow) * 64 + kcw * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wd, (__attribute__((address_space(3))) void*)(smem + OFF_WD + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = ty0 + 2 * uw + i, x = tx0 + lr;
      const unsigned xrow = (y < p.H && x < p.W) ? (unsigned)((img0 + (long long)y * p.W + x) * 128) : OOB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xfr[i][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(xrow + (unsigned)((ks * 4 + lq) * 16)), 0, 0);
    }
  }
  {
    char* t2 = smem + OFF_T2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o_ = (2 * uw + i) * 16 + lr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = jp * 32 + lq * 8;
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][2 * jp + q][r] * sc[r] + sh[r], 0.f);
        }
        *reinterpret_cast<bf16x8*>(t2 + swz(o_, jp * 4 + lq)) = o;
      }
    }
  }
  __syncthreads();

  BSTAMP(5);
  // ------------------------------------------------------------------ phase 3: y = relu(bn3(conv3(t2)) + res), 2 x 128 output channels
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if constexpr (DS) {
      // the residual of this half: bn_d(conv_d(x)) rounded to bf16 -- the bits the separate launch would have stored
      f32x4 accd[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) accd[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + OFF_WD + wswz(half * 128 + wrow(j, lr), ks * 4 + lq));
#pragma unroll
          for (int i = 0; i < 2; ++i) accd[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xfr[i][ks]), accd[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const int c = half * 128 + jp * 32 + lq * 8;
          bf16x8 o;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vsd + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vbd + c + 4 * h2);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * h2 + r] = (bf16_t)(accd[i][2 * jp + h2][r] * sc[r] + sh[r]);
          }
          rv[half][i][jp] = __builtin_bit_cast(u32x4_t, o);
        }
    }
    f32x4 acc3[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* t2 = smem + OFF_T2;
    const char* ws = smem + OFF_W3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t2 + swz((2 * uw + i) * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + wswz(half * 128 + wrow(j, lr), ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[i], acc3[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const int c = half * 128 + jp * 32 + lq * 8;
        const u32x4_t q = rv[half][i][jp];
        bf16x8 o;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc3[i][2 * jp + h2][r] * sc[r] + sh[r];
          v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
          v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * h2 + r] = (bf16_t)fmaxf(v[r], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_y, (int)(prow[i] + (unsigned)(c * 2)), 0, 0);
      }
  }
  BSTAMP(6);
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BSTAMP(7);
#endif
}

}  // namespace

#ifdef AOD_TILE_TIMING
extern "C" int aod_dbg_set_bn_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bn_stamps), &buf, sizeof(buf)); }
#endif

static int launch_bnk(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                      const float* b2, const void* w3, const float* s3, const float* b3, const void* res, const void* wd, const float* sd,
                      const float* bd, void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck64: null pointer");
  AOD_CHECK_ARG(Cin >= 64 && Cin % 64 == 0 && B >= 1 && H >= 1 && W >= 1, "bottleneck64: Cin %d must be a multiple of 64", Cin);
  AOD_CHECK_ARG((long long)B * H * W * (Cin > 256 ? Cin : 256) * 2 < 0xe0000000ll, "bottleneck64: operand larger than 3.5 GiB");
  BnkArgs a;
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.res = (const bf16_t*)res; a.y = (bf16_t*)y;
  a.wd = (const bf16_t*)wd; a.sd = sd; a.bd = bd;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  if (wd) hipLaunchKernelGGL(bottleneck64_fwd_kernel<true>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bottleneck64_fwd_kernel<false>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_bottleneck64_fwd(const void* x, int Cin, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                    const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* res,
                                    void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(res, "bottleneck64: null pointer");
  return launch_bnk(x, Cin, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, res, nullptr, nullptr, nullptr, y, stream);
}

extern "C" int aod_bottleneck64_ds_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                       const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd,
                                       const float* sd, const float* bd, void* y, aod_stream_t stream) {
  AOD_CHECK_ARG(wd && sd && bd, "bottleneck64_ds: null pointer");
  return launch_bnk(x, 64, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, nullptr, wd, sd, bd, y, stream);
}
