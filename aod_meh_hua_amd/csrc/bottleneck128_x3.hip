// Whole ResNet IDENTITY bottleneck of the 128-plane stage (layer2: 512 -> 128 -> 128 -> 512 channels) in ONE kernel, REFERENCE-PRECISION
// mode (conv.hip "X3"): the x3 twin of bottleneck_wide.hip's 128-plane kernel
//   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + x)                (mmdet/models/backbones/resnet.py:262-301)
// on X-layout rows (bf16 head + tail pairs, [h32 | l32] per 32 channels), every product as xh*wh + xl*wh + xh*wl in one fp32 accumulator.
// As three launches the block moves 8 KB per pixel through HBM (x for conv1, x again as the residual, y, the two 128-channel intermediates
// written and re-read: 524 MB at 16 x 64 x 64 pixels) in 185 us; fused, x and y cross HBM once (+ the two intermediates once, written, when
// the block is the forward of a training step).
//
// One 8-wave workgroup = one 8 x 16 pixel tile of one image (waves 4 x 2 over pixel row pairs x output-channel halves):
//   phase 1  conv1 on the 10 x 18 HALO (180 pixels, 12 blocks of 16): x and the X filter streamed through LDS in 64-column K-steps (= 32
//            channels: heads in k-block 0, tails in k-block 1) by LDS-DMA, two stages under counted waits; t1 = relu(bn1(.)) -> LDS as X
//            rows (four sub-images of 32 channels, 180 rows), ZERO outside the image (conv2 pads t1);
//   phase 2  conv2 as 36 (channel group, tap) K-steps, TAP INNERMOST -- the order conv_igemm_kernel<X3> walks a 256-column layer in --:
//            pixel fragments gathered from t1 in LDS, 16-KB filter slices through a 4-slot ring (three in flight, one counted s_waitcnt
//            vmcnt + one raw s_barrier per step); t2 = relu(bn2(.)) -> LDS over t1;
//   phase 3  conv3 as 4 chunks of 128 output channels x 4 channel groups = 16 more slices of the SAME ring; per chunk the residual pieces
//            (head + tail, 16 B each) are requested at the chunk's first step and the epilogue runs in registers: + residual, ReLU, head /
//            tail stores.
// Optional outputs t1 / t2 (interior pixels, X rows): with them the launch is the forward of a training step -- the three convs are then
// recorded as autograd nodes around its outputs (functional.conv_bn_act(pre=...)) and the backward pass is unchanged.
// Products are computed transposed (filter fragment = the MFMA's A operand) with the paired-block row permutation of bottleneck.hip, so a
// lane holds 8 consecutive channels of one pixel and every intermediate / residual / output piece is 16 B.  Same products in the same order
// per accumulator, same fp32 epilogue arithmetic and the same head / tail roundings as the three conv_igemm_kernel<X3> launches: identical bits.
#include "common.h"

namespace {

struct Bw3Args {
  const bf16_t* x;       // X rows [B*H*W][1024]
  const bf16_t* w1;      // X filter [128][1024]
  const bf16_t* w2;      // X filter [128][9][256]
  const bf16_t* w3;      // X filter [512][256]
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  bf16_t* y;             // X rows [B*H*W][1024]
  bf16_t* t1;            // X rows [B*H*W][256] or null
  bf16_t* t2;            // X rows [B*H*W][256] or null
  int B, H, W, tiles_y, tiles_x;
};

constexpr int P = 128, XC = 1024, XP = 256;                              // planes; X-layout widths of x / y and of t1 / t2
constexpr int TH = 8, TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * HW_;     // 180 halo pixels
constexpr int HROWS = 192;                                               // the conv1 tile: 12 row blocks of 16
constexpr int XCH = HROWS * 128, W1CH = P * 128, STG1 = XCH + W1CH;      // one K-step of the halo (24 576) + of the conv1 filter (16 384)
constexpr int SLOT = P * 128, NSLOT = 4;                                 // filter ring: 4 slots of [128 rows][128 B] at offset 0
constexpr int OFF_T1 = NSLOT * SLOT;                                     // 65 536: t1 as 4 sub-images [180][128 B]; later t2 [4][128][128 B]
constexpr int T1SUB = HPIX * 128, T2SUB = TH * TW * 128;
constexpr int OFF_VEC = OFF_T1 + 4 * T1SUB;                              // 157 696: s1 b1 s2 b2 [128] | s3 [512] | b3 [512] fp32
constexpr int LDS_BYTES = OFF_VEC + (4 * P + 2 * 4 * P) * 4;             // 163 840 = all of the CU's LDS
constexpr int NS2 = 36, NS = NS2 + 16;                                   // ring steps: conv2, then conv3
static_assert(2 * STG1 <= OFF_VEC && 4 * T2SUB <= 4 * T1SUB && LDS_BYTES <= 160 * 1024, "LDS map");
constexpr unsigned OOB = 0xf0000000u;

#define AOD_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    AOD_VMCASE(0) AOD_VMCASE(4) AOD_VMCASE(5) AOD_VMCASE(6) AOD_VMCASE(8) AOD_VMCASE(12) AOD_VMCASE(16) AOD_VMCASE(18) AOD_VMCASE(20)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef AOD_VMCASE

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { h[j] = (bf16_t)v[j]; l[j] = (bf16_t)(v[j] - (float)h[j]); }
}

// Vector-memory operations of one wave in phases 2 / 3, in issue order (S = a ring slice, 2 instructions; the optional outputs are ALWAYS
// issued -- through an empty descriptor when the tensor is not wanted -- so the counts hold for every launch):
//   S0 S1 S2 | t1 stores x12 | step s: [wait S_s] barrier, S_{s+3} ...                                        (conv2: s = 0 .. 35)
//   t2 stores x8 | step s: [wait S_s] barrier, S_{s+3}, (first step of a chunk: 8 residual loads) ... (last step of a chunk: [wait the
//   residual loads] epilogue, 8 output stores)                                                                (conv3: s = 36 .. 51)
// WAIT[s] = operations younger than slice s at its wait (they may stay in flight); RESW = younger than a chunk's residual loads at its epilogue.
__device__ __forceinline__ int wait_of(int s) {
  if (s < 3) return 16;
  if (s < NS2) return 4;
  if (s == NS - 1) return 8;
  if (s == NS - 2) return 18;
  const int kg = (s - NS2) & 3;
  return (kg == 0 || kg == 3) ? 12 : 20;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck128x3_fwd_kernel(const Bw3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wp = uw >> 1, wc = uw & 1;                    // wave grid: 4 (pixel row pairs) x 2 (output-channel halves)
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const long long npix = (long long)p.B * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(npix * XC * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, P * XC * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, P * 9 * XP * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, 4 * P * XP * 2, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(npix * XC * 2), 0x00020000);
  // the optional outputs through descriptors that are EMPTY when the tensor is not wanted: the stores are always issued (the counted waits
  // rely on that) and dropped by the range check
  const auto rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1, 0, p.t1 ? (int)(npix * XP * 2) : 0, 0x00020000);
  const auto rsrc_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t2, 0, p.t2 ? (int)(npix * XP * 2) : 0, 0x00020000);

  // LDS-DMA lane roles (bottleneck.hip): one wave-instruction fills 8 rows x 8 slots; lane -> row (lane >> 3) of its group, slot lane & 7,
  // source chunk slot ^ key(row); a wave serves row groups uw + 8 i, for which the keys are the same
  const int drow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int kcw = kc ^ (((uw >> 1) & 1) << 1);

  auto halo_pix = [&](int h, int& y, int& x) -> bool {
    const int hy = h / HW_, hx = h - hy * HW_;
    y = ty0 - 1 + hy; x = tx0 - 1 + hx;
    return h < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
  };

  // folded BN vectors -> LDS once (an ordinary global load beside LDS-DMA makes hipcc drain the DMA queue where the value is used)
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  {
    if (t < 128) vec[t] = p.s1[t];
    else if (t < 256) vec[t] = p.b1[t - 128];
    else if (t < 384) vec[t] = p.s2[t - 256];
    else vec[t] = p.b2[t - 384];
    vec[512 + t] = p.s3[t];
    vec[1024 + t] = p.b3[t];
  }
  const float* const vs1 = vec, * const vb1 = vec + 128, * const vs2 = vec + 256, * const vb2 = vec + 384, * const vs3 = vec + 512, * const vb3 = vec + 1024;

  // ------------------------------------------------------------------ phase 1: t1 = relu(bn1(conv1(x))) on the halo
  unsigned xoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int h = 8 * (uw + 8 * i) + drow;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * XC + kc * 8) * 2) : OOB;
  }
  unsigned w1off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w1off[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * XC + kcw * 8) * 2);
  auto issue1 = [&](int buf) {                   // 3 x-halo instructions + 2 filter instructions per wave and stage
    char* xs = smem + buf * STG1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned off = xoff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(xs + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      xoff[i] += 128;                                  // (an OOB row stays out of range)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = w1off[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(xs + XCH + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      w1off[i] += 128;
    }
  };
  f32x4 acc1[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NK1 = XC / 64;
  issue1(0);
  issue1(1);
  for (int kt = 0; kt < NK1; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < NK1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (kt == 0: this thread's vec[] stores)
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage kt has landed
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = smem + buf * STG1;
    const char* ws = xs + XCH;
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), lq));
      wl[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), 4 + lq));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = (3 * wp + i) * 16 + lr;
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xs + swz(row, lq)), xl = *reinterpret_cast<const bf16x8*>(xs + swz(row, 4 + lq));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (the order of conv_igemm_kernel<X3>: heads x heads, activation tails x filter heads, activation heads x filter tails)
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh, acc1[i][j], 0, 0, 0);
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xl, acc1[i][j], 0, 0, 0);
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh, acc1[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading stage kt: its buffer can be refilled
    if (kt + 2 < NK1) issue1(buf);
  }

  // the stages are dead: the first three filter slices stream into the ring under epilogue 1
  unsigned w2lane[2], w3lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    w2lane[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * 9 * XP + kcw * 8) * 2);
    w3lane[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * XP + kcw * 8) * 2);
  }
  auto issue_slice = [&](int s) {                // s: ring step (wave-uniform); conv2 step = channel group * 9 + tap, conv3 step = chunk * 4 + group
    char* dst = smem + (s & (NSLOT - 1)) * SLOT;
    if (s < NS2) {
      const int cg = s / 9, tap = s - cg * 9;
      const unsigned koff = (unsigned)((tap * XP + cg * 64) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(dst + (uw + 8 * i) * 1024), 16, w2lane[i] + koff, 0, 0, 0);
    } else {
      const int q = s - NS2, n3 = q >> 2, kg = q & 3;
      const unsigned koff = (unsigned)((n3 * P * XP + kg * 64) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(dst + (uw + 8 * i) * 1024), 16, w3lane[i] + koff, 0, 0, 0);
    }
  };
  issue_slice(0);
  issue_slice(1);
  issue_slice(2);
  // epilogue 1: t1 -> LDS (zero outside the image) and, for the tile's own pixels, -> global (training forward)
  {
    char* t1 = smem + OFF_T1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int h = (3 * wp + i) * 16 + lr;
      int y, x;
      const bool ok = halo_pix(h, y, x);
      const int hy = h / HW_, hx = h - hy * HW_;
      const bool inner = ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (XP * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = 64 * wc + 32 * jp + lq * 8;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * q + r] = ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f;
        }
        bf16x8 oh, ol;
        split8(v, oh, ol);
        if (h < HPIX) {
          *reinterpret_cast<bf16x8*>(t1 + (2 * wc + jp) * T1SUB + swz(h, lq)) = oh;
          *reinterpret_cast<bf16x8*>(t1 + (2 * wc + jp) * T1SUB + swz(h, 4 + lq)) = ol;
        }
        const unsigned col = (unsigned)((2 * wc + jp) * 128 + lq * 16);      // byte offset of the 8 heads of channel group 2 wc + jp in an X row
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), rsrc_t1, (int)(grow + col), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), rsrc_t1, (int)(grow + col + 64u), 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ phase 2: t2 = relu(bn2(conv2(t1))), filter slices through the ring
  f32x4 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    int cg = 0, tap = 0;
    for (int s = 0; s < NS2; ++s) {
      wait_vm_dyn(wait_of(s));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (s == 0: this wave's t1 pieces are written)
      __builtin_amdgcn_s_barrier();               // slice s complete; every wave is done with step s - 1 (its slot is free)
      __builtin_amdgcn_sched_barrier(0);
      issue_slice(s + 3);                         // (always <= 51 here)
      const int r = tap / 3, q = tap - r * 3;
      const char* t1 = smem + OFF_T1 + cg * T1SUB;
      const char* ws = smem + (s & (NSLOT - 1)) * SLOT;
      bf16x8 wh[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), lq));
        wl[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), 4 + lq));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (2 * wp + i + r) * HW_ + lr + q;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(t1 + swz(row, lq)), al = *reinterpret_cast<const bf16x8*>(t1 + swz(row, 4 + lq));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc2[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (++tap == 9) { tap = 0; ++cg; }
    }
  }
  __builtin_amdgcn_s_barrier();                   // every wave is done reading t1: t2 goes over it
  __builtin_amdgcn_sched_barrier(0);
  unsigned prow[2];                               // byte offset of the lane's pixel rows in a [B*H*W][1024] X tensor
  {
    char* t2 = smem + OFF_T1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = ty0 + 2 * wp + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      prow[i] = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (XC * 2)) : OOB;
      const unsigned grow = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (XP * 2)) : OOB;
      const int o_ = (2 * wp + i) * 16 + lr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = 64 * wc + 32 * jp + lq * 8;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * q + r] = fmaxf(acc2[i][2 * jp + q][r] * sc[r] + sh[r], 0.f);
        }
        bf16x8 oh, ol;
        split8(v, oh, ol);
        *reinterpret_cast<bf16x8*>(t2 + (2 * wc + jp) * T2SUB + swz(o_, lq)) = oh;
        *reinterpret_cast<bf16x8*>(t2 + (2 * wc + jp) * T2SUB + swz(o_, 4 + lq)) = ol;
        const unsigned col = (unsigned)((2 * wc + jp) * 128 + lq * 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), rsrc_t2, (int)(grow + col), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), rsrc_t2, (int)(grow + col + 64u), 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ phase 3: y = relu(bn3(conv3(t2)) + x), 4 chunks of 128 output channels
#pragma unroll
  for (int n3 = 0; n3 < 4; ++n3) {
    f32x4 acc3[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4_t rh[2][2], rl[2][2];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int s = NS2 + n3 * 4 + kg;
      wait_vm_dyn(wait_of(s));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (first step: this wave's t2 pieces are written)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + 3 < NS) issue_slice(s + 3);
      if (kg == 0) {
        // residual pieces of this chunk (x was read by this XCD for conv1: L2 hits); 16-B heads and tails of the lane's 8 channels
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const unsigned col = (unsigned)((n3 * 4 + 2 * wc + jp) * 128 + lq * 16);
            rh[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(prow[i] + col), 0, 0);
            rl[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(prow[i] + col + 64u), 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      const char* t2 = smem + OFF_T1 + kg * T2SUB;
      const char* ws = smem + (s & (NSLOT - 1)) * SLOT;
      bf16x8 wh[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), lq));
        wl[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), 4 + lq));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (2 * wp + i) * 16 + lr;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(t2 + swz(row, lq)), al = *reinterpret_cast<const bf16x8*>(t2 + swz(row, 4 + lq));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc3[i][j], 0, 0, 0);
          acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc3[i][j], 0, 0, 0);
          acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc3[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the chunk's residual loads have landed when at most the slices issued since (3, or none behind the last chunk) are in flight
    if (n3 < 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = n3 * 128 + 64 * wc + 32 * jp + lq * 8;
        const u32x4_t qh = rh[i][jp], ql = rl[i][jp];
        float v[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * h2 + r] = acc3[i][2 * jp + h2][r] * sc[r] + sh[r];
          v[4 * h2 + 0] += __uint_as_float(qh[2 * h2] << 16) + __uint_as_float(ql[2 * h2] << 16);
          v[4 * h2 + 1] += __uint_as_float(qh[2 * h2] & 0xffff0000u) + __uint_as_float(ql[2 * h2] & 0xffff0000u);
          v[4 * h2 + 2] += __uint_as_float(qh[2 * h2 + 1] << 16) + __uint_as_float(ql[2 * h2 + 1] << 16);
          v[4 * h2 + 3] += __uint_as_float(qh[2 * h2 + 1] & 0xffff0000u) + __uint_as_float(ql[2 * h2 + 1] & 0xffff0000u);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
        bf16x8 oh, ol;
        split8(v, oh, ol);
        const unsigned col = (unsigned)((n3 * 4 + 2 * wc + jp) * 128 + lq * 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), rsrc_y, (int)(prow[i] + col), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), rsrc_y, (int)(prow[i] + col + 64u), 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

extern "C" int aod_bottleneck128x3_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                       const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                       void* t2, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck128x3: null pointer");
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck128x3: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * XC * 2 < 0xe0000000ll, "bottleneck128x3: operand larger than 3.5 GiB");
  AOD_CHECK_ARG(x != y, "bottleneck128x3: y must not alias x (the residual is read after neighbouring tiles have stored)");
  Bw3Args a;
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2;
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck128x3_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL(bottleneck128x3_fwd_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}
