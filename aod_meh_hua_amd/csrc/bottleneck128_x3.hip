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
  // backward form (BWD instance): ReLU masks = the block's saved activations (X rows: the head decides the sign), fp32 column sums of the results
  const bf16_t* m1;      // [B*H*W][256]    mask of the first product  (forward t2)
  const bf16_t* m2;      // [B*H*W][256]    mask of the second product (forward t1)
  const bf16_t* m3;      // [B*H*W][1024]   mask of the third product  (forward x)
  float* cs1; float* cs2; float* cs3;     // [128] [128] [512], += (atomics)
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
static_assert(3 * STG1 <= OFF_VEC && 4 * T2SUB <= 4 * T1SUB && LDS_BYTES <= 160 * 1024, "LDS map");
constexpr unsigned OOB = 0xf0000000u;

#ifdef AOD_TILE_TIMING
// debug build only (tools/dbg/b128x3_timing.py): per-workgroup wall-clock stamps (100 MHz) at the phase boundaries
__device__ unsigned long long* g_bw3_stamps = nullptr;
#define WSTAMP(k) do { if (g_bw3_stamps && threadIdx.x == 0) g_bw3_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define TICK() __builtin_amdgcn_s_memtime()
#define TACC(acc, a, b) acc += (b) - (a)
#else
#define WSTAMP(k) do {} while (0)
#define TICK() 0ull
#define TACC(acc, a, b) do {} while (0)
#endif

#define AOD_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    AOD_VMCASE(0) AOD_VMCASE(4) AOD_VMCASE(5) AOD_VMCASE(6) AOD_VMCASE(8) AOD_VMCASE(12) AOD_VMCASE(16) AOD_VMCASE(18) AOD_VMCASE(20) AOD_VMCASE(22) AOD_VMCASE(24)
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
// BWD: 4 mask loads of the second epilogue behind the t1 stores, and 12 loads (8 residual + 4 mask pieces) per chunk instead of 8.
// (pipelined form: step s waits for slice s + 1 -- four slices go out ahead of the loop, step s issues slice s + 4)
template <bool BWD>
__device__ __forceinline__ int wait_pro() { return 6 + 12 + (BWD ? 4 : 0); }
template <bool BWD>
__device__ __forceinline__ int wait_of(int s) {
  constexpr int R = BWD ? 12 : 8, MK2 = BWD ? 4 : 0;
  if (s < 3) return 16 + MK2;
  if (s < NS2) return 4;
  if (s == NS - 2) return 8 + R;                  // (slice 51: the last chunk's residual loads and the stores of the chunk before are younger)
  if (s == NS - 3) return 10 + R;
  const int kg = (s - NS2) & 3;
  return kg == 0 ? 12 : (kg == 3 ? 4 + R : 12 + R);
}

// bit j of the result: element j of the 16-B piece (8 bf16 heads) is > 0 -- the ReLU mask of 8 channels in 8 bits
__device__ __forceinline__ unsigned pos_bits(const u32x4_t q) {
  unsigned m = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    m |= (__uint_as_float(q[w] << 16) > 0.f ? 1u : 0u) << (2 * w);
    m |= (__uint_as_float(q[w] & 0xffff0000u) > 0.f ? 1u : 0u) << (2 * w + 1);
  }
  return m;
}
// sum over the 16 lanes that share lq (one DPP row): four rotate-and-add steps on the vector ALU, every lane ends up with the total
template <int N> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum_lr(float v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}

// BWD: the same three products are the block's DGRAD chain (csrc/conv.hip dgrad epilogue semantics: result = mask > 0 ? acc (+ res) : 0,
// fp32 column sums of the results): x = the finished gradient G of the block output, w1 / w2 / w3 = the scale-folded X dgrad filters of
// conv3 / conv2 / conv1, t1 / t2 / y = the gradients of forward t2 / t1 / x, residual = G (the skip branch).
template <bool BWD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck128x3_kernel(const Bw3Args p) {
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
  // BWD: the column sums of the three results are collected in LDS (ds_add_f32) and leave the workgroup as ONE global atomic per channel
  float* const csl = vec;                          // [0, 512): third result, [512, 640): first, [640, 768): second
  if constexpr (BWD) {
    vec[t] = 0.f;
    if (t < 2 * P) vec[512 + t] = 0.f;
  } else {
    if (t < 128) vec[t] = p.s1[t];
    else if (t < 256) vec[t] = p.b1[t - 128];
    else if (t < 384) vec[t] = p.s2[t - 256];
    else vec[t] = p.b2[t - 384];
    vec[512 + t] = p.s3[t];
    vec[1024 + t] = p.b3[t];
  }
  const float* const vs1 = vec, * const vb1 = vec + 128, * const vs2 = vec + 256, * const vb2 = vec + 384, * const vs3 = vec + 512, * const vb3 = vec + 1024;

  WSTAMP(0);
  // ------------------------------------------------------------------ phase 1: t1 = relu(bn1(conv1(x))) on the halo
  const auto rsrc_m1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m1, 0, BWD ? (int)(npix * XP * 2) : 0, 0x00020000);
  const auto rsrc_m2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m2, 0, BWD ? (int)(npix * XP * 2) : 0, 0x00020000);
  const auto rsrc_m3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m3, 0, BWD ? (int)(npix * XC * 2) : 0, 0x00020000);
  u32x4_t mk1[3][2];                               // BWD: head pieces of the masks of this lane's halo pixels (out-of-image pixels read 0 = masked)
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int h = (3 * wp + i) * 16 + lr;
      int y, x;
      const unsigned hrow = halo_pix(h, y, x) ? (unsigned)((img0 + (long long)y * p.W + x) * (XP * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) mk1[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m1, (int)(hrow + (unsigned)((2 * wc + jp) * 128 + lq * 16)), 0, 0);
    }
  }
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
  // (the K offset of a stage rides in the instruction's scalar offset: no per-stage vector arithmetic.  The loops of all three phases are
  // fully unrolled and every LDS fragment address is a per-lane constant from a table built once + an immediate: with run-time step indices
  // the ~70 vector instructions of address arithmetic per wave and step outran the issue slots the MFMAs leave -- 1 400 cycles per conv2 step
  // for 768 cycles of matrix work, tools/dbg/b128x3_timing.py)
  auto issue1 = [&](int buf, int kt) {            // 3 x-halo instructions + 2 filter instructions per wave and stage
    char* xs = smem + buf * STG1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned off = xoff[i];               // (an OOB row stays out of range)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(xs + (uw + 8 * i) * 1024), 16, off, kt * 128, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = w1off[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(xs + XCH + (uw + 8 * i) * 1024), 16, off, kt * 128, 0, 0);
    }
  };
  // per-lane fragment addresses (heads; the tails sit 64 B further: chunk 4 + lq = the head's slot ^ 4)
  unsigned wfrag[4], xfrag1[3], afrag2[2][9], afrag3[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) wfrag[j] = (unsigned)wswz(wrow(4 * wc + j, lr), lq);
#pragma unroll
  for (int i = 0; i < 3; ++i) xfrag1[i] = (unsigned)swz((3 * wp + i) * 16 + lr, lq);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r0_ = tap / 3, q0_ = tap - r0_ * 3;
      const int r = BWD ? 2 - r0_ : r0_, q = BWD ? 2 - q0_ : q0_;      // dgrad: tap (r, s) of the [I][R][S][O] pack reads pixel (y + 1 - r, x + 1 - s)
      afrag2[i][tap] = (unsigned)(OFF_T1 + swz((2 * wp + i + r) * HW_ + lr + q, lq));
    }
    afrag3[i] = (unsigned)(OFF_T1 + swz((2 * wp + i) * 16 + lr, lq));
  }
  auto lds16 = [&](unsigned a) { return *reinterpret_cast<const bf16x8*>(smem + a); };
  f32x4 acc1[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // K loop, software-pipelined: THREE stages (the t1 area is dead until the epilogue), the fragments of K-step kt + 1 are read into a second
  // register set while the MFMAs of step kt run -- with one set the eight waves read in a burst behind every barrier and the matrix pipe
  // waited for it (2 225 cycles per step against 1 152 of MFMA work; tools/dbg/b128x3_timing.py)
  constexpr int NK1 = XC / 64;
  struct F1 { bf16x8 wh[4], wl[4], xh[3], xl[3]; };
  auto read1 = [&](int buf, F1& f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.wh[j] = lds16(buf * STG1 + XCH + wfrag[j]);
      f.wl[j] = lds16(buf * STG1 + XCH + (wfrag[j] ^ 64u));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f.xh[i] = lds16(buf * STG1 + xfrag1[i]);
      f.xl[i] = lds16(buf * STG1 + (xfrag1[i] ^ 64u));
    }
  };
  auto mma1 = [&](const F1& f) {
    // per accumulator the order of conv_igemm_kernel<X3> -- heads x heads, activation tails x filter heads, activation heads x filter tails --,
    // product-major over the accumulators so that no MFMA waits for the one before it
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[j], f.xh[i], acc1[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[j], f.xl[i], acc1[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[j], f.xh[i], acc1[i][j], 0, 0, 0);
  };
  // one pipeline step: stage kt + 1 has landed for every wave and every wave holds the fragments of stage kt in registers (so its buffer
  // takes stage kt + 3); then the reads of kt + 1 go out and the MFMAs of kt run beside them
  auto step1 = [&](int kt, const F1& cur, F1& nxt) {
    if (kt + 1 < NK1) {
      if (kt + 2 < NK1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of stage kt are complete
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 3 < NK1) issue1(kt % 3, kt + 3);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < NK1) read1((kt + 1) % 3, nxt);
    mma1(cur);
    __builtin_amdgcn_sched_barrier(0);
  };
  issue1(0, 0);
  issue1(1, 1);
  issue1(2, 2);
  F1 fa, fb;
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (this thread's vec[] stores)
  __builtin_amdgcn_s_barrier();                   // every wave's part of stage 0 has landed
  __builtin_amdgcn_sched_barrier(0);
  read1(0, fa);
#pragma unroll
  for (int kt = 0; kt < NK1; kt += 2) {
    step1(kt, fa, fb);
    step1(kt + 1, fb, fa);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // every wave is done with the stages

  WSTAMP(1);
  // the stages are dead: the first four filter slices stream into the ring under epilogue 1
  unsigned w2lane[2], w3lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    w2lane[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * 9 * XP + kcw * 8) * 2);
    w3lane[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * XP + kcw * 8) * 2);
  }
  auto issue_slice = [&](int s) {                // s: ring step (compile-time after unrolling); conv2 step = channel group * 9 + tap, conv3 step = chunk * 4 + group
    char* dst = smem + (s & (NSLOT - 1)) * SLOT;
    if (s < NS2) {
      const int cg = s / 9, tap = s - cg * 9;
      const int koff = (tap * XP + cg * 64) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(dst + (uw + 8 * i) * 1024), 16, w2lane[i], koff, 0, 0);
    } else {
      const int q = s - NS2, n3 = q >> 2, kg = q & 3;
      const int koff = (n3 * P * XP + kg * 64) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(dst + (uw + 8 * i) * 1024), 16, w3lane[i], koff, 0, 0);
    }
  };
  issue_slice(0);
  issue_slice(1);
  issue_slice(2);
  issue_slice(3);
  __builtin_amdgcn_sched_barrier(0);
  // epilogue 1: t1 -> LDS (zero outside the image) and, for the tile's own pixels, -> global (training forward)
  u32x4_t mk2[2][2];                              // BWD: mask pieces of the second epilogue (forward t1 at the tile's own pixels)
  {
    char* t1 = smem + OFF_T1;
    float cs1v[2][8] = {};
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
        if constexpr (BWD) {
          const unsigned mb = pos_bits(mk1[i][jp]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[4 * q + r] = ((mb >> (4 * q + r)) & 1u) ? acc1[i][2 * jp + q][r] + 0.f : 0.f;
              cs1v[jp][4 * q + r] += inner ? v[4 * q + r] : 0.f;
            }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f;
          }
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
    if constexpr (BWD) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = ty0 + 2 * wp + i, x = tx0 + lr;
        const unsigned r2 = (y < p.H && x < p.W) ? (unsigned)((img0 + (long long)y * p.W + x) * (XP * 2)) : OOB;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) mk2[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m2, (int)(r2 + (unsigned)((2 * wc + jp) * 128 + lq * 16)), 0, 0);
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = sum_lr(cs1v[jp][j]);
          if (lr == 0) atomicAdd(csl + 4 * P + 64 * wc + 32 * jp + lq * 8 + j, v);
        }
    }
  }

  WSTAMP(2);
  // ------------------------------------------------------------------ phase 2: t2 = relu(bn2(conv2(t1))), filter slices through the ring
  f32x4 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Ring steps, software-pipelined like phase 1: at step s every wave holds the fragments of slice s in registers -- slot s % 4 takes slice
  // s + 4 --, slice s + 1 has landed, its fragments (and the pixel fragments of step s + 1) are read while the MFMAs of step s run.
  struct F2 { bf16x8 wh[4], wl[4], ah[2], al[2]; };
  auto read_w = [&](int s, F2& f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.wh[j] = lds16((s & (NSLOT - 1)) * SLOT + wfrag[j]);
      f.wl[j] = lds16((s & (NSLOT - 1)) * SLOT + (wfrag[j] ^ 64u));
    }
  };
  auto read2 = [&](int s, F2& f) {                // conv2 step s = channel group * 9 + tap
    read_w(s, f);
    const int cg = s / 9, tap = s - cg * 9;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.ah[i] = lds16(cg * T1SUB + afrag2[i][tap]);
      f.al[i] = lds16(cg * T1SUB + (afrag2[i][tap] ^ 64u));
    }
  };
  auto mma2 = [&](const F2& f, f32x4 (&acc)[2][4]) {      // (product-major: see mma1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[j], f.al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[j], f.ah[i], acc[i][j], 0, 0, 0);
  };
  [[maybe_unused]] unsigned long long tk_wait = 0, tk_lds = 0, tk_bar = 0, tk_work = 0;
  auto step2 = [&](int s, const F2& cur, F2& nxt) {
    [[maybe_unused]] const unsigned long long c0 = TICK();
    wait_vm_dyn(wait_of<BWD>(s));                 // slice s + 1 has landed (this wave's part)
    [[maybe_unused]] const unsigned long long c1 = TICK();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of slice s are complete
    [[maybe_unused]] const unsigned long long c2 = TICK();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const unsigned long long c3 = TICK();
    issue_slice(s + 4);                           // (always <= 51 here)
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < NS2) read2(s + 1, nxt);           // (the pixel fragments of step 36 do not exist yet: t2 is this phase's result)
    mma2(cur, acc2);
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const unsigned long long c4 = TICK();
    TACC(tk_wait, c0, c1); TACC(tk_lds, c1, c2); TACC(tk_bar, c2, c3); TACC(tk_work, c3, c4);
  };
  F2 ga, gb;
  wait_vm_dyn(wait_pro<BWD>());                   // slice 0 (younger: slices 1 - 3, the t1 stores, BWD: the mask loads)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's t1 pieces are written
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read2(0, ga);
#pragma unroll
  for (int s = 0; s < NS2; s += 2) {
    step2(s, ga, gb);
    step2(s + 1, gb, ga);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // every wave is done reading t1: t2 goes over it
  __builtin_amdgcn_sched_barrier(0);
  WSTAMP(3);
  unsigned prow[2];                               // byte offset of the lane's pixel rows in a [B*H*W][1024] X tensor
  {
    char* t2 = smem + OFF_T1;
    float cs2v[2][8] = {};
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
        if constexpr (BWD) {
          const unsigned mb = pos_bits(mk2[i][jp]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[4 * q + r] = ((mb >> (4 * q + r)) & 1u) ? acc2[i][2 * jp + q][r] + 0.f : 0.f;
              cs2v[jp][4 * q + r] += v[4 * q + r];
            }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = fmaxf(acc2[i][2 * jp + q][r] * sc[r] + sh[r], 0.f);
          }
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
    if constexpr (BWD) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = sum_lr(cs2v[jp][j]);
          if (lr == 0) atomicAdd(csl + 5 * P + 64 * wc + 32 * jp + lq * 8 + j, v);
        }
    }
  }

  WSTAMP(4);
  // ------------------------------------------------------------------ phase 3: y = relu(bn3(conv3(t2)) + x), 4 chunks of 128 output channels
  auto read3 = [&](int s, F2& f) {                // conv3 step s = 36 + chunk * 4 + channel group
    read_w(s, f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.ah[i] = lds16(((s - NS2) & 3) * T2SUB + afrag3[i]);
      f.al[i] = lds16(((s - NS2) & 3) * T2SUB + (afrag3[i] ^ 64u));
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's t2 pieces are written
  __builtin_amdgcn_s_barrier();                   // t2 complete (slice 36 was waited for at step 35)
  __builtin_amdgcn_sched_barrier(0);
  read3(NS2, ga);
#pragma unroll
  for (int n3 = 0; n3 < 4; ++n3) {
    f32x4 acc3[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4_t rh[2][2], rl[2][2], mk3[2][2];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int s = NS2 + n3 * 4 + kg;
      if (s + 1 < NS) wait_vm_dyn(wait_of<BWD>(s));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + 4 < NS) issue_slice(s + 4);
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
        if constexpr (BWD) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
              mk3[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m3, (int)(prow[i] + (unsigned)((n3 * 4 + 2 * wc + jp) * 128 + lq * 16)), 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < NS) read3(s + 1, (kg & 1) ? ga : gb);
      mma2((kg & 1) ? gb : ga, acc3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the chunk's residual (and mask) loads have landed when at most the slices issued since (3, or none behind the last chunk) are in flight
    if (n3 < 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float c3v[2][8] = {};                         // BWD: this lane's column sums of the chunk (its two pixel rows)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = n3 * 128 + 64 * wc + 32 * jp + lq * 8;
        const u32x4_t qh = rh[i][jp], ql = rl[i][jp];
        float v[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          if constexpr (BWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * h2 + r] = acc3[i][2 * jp + h2][r] + 0.f;
          } else {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * h2 + r] = acc3[i][2 * jp + h2][r] * sc[r] + sh[r];
          }
          v[4 * h2 + 0] += __uint_as_float(qh[2 * h2] << 16) + __uint_as_float(ql[2 * h2] << 16);
          v[4 * h2 + 1] += __uint_as_float(qh[2 * h2] & 0xffff0000u) + __uint_as_float(ql[2 * h2] & 0xffff0000u);
          v[4 * h2 + 2] += __uint_as_float(qh[2 * h2 + 1] << 16) + __uint_as_float(ql[2 * h2 + 1] << 16);
          v[4 * h2 + 3] += __uint_as_float(qh[2 * h2 + 1] & 0xffff0000u) + __uint_as_float(ql[2 * h2 + 1] & 0xffff0000u);
        }
        if constexpr (BWD) {
          const unsigned mb = pos_bits(mk3[i][jp]);
#pragma unroll
          for (int r = 0; r < 8; ++r) { v[r] = ((mb >> r) & 1u) ? v[r] : 0.f; c3v[jp][r] += v[r]; }
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        bf16x8 oh, ol;
        split8(v, oh, ol);
        const unsigned col = (unsigned)((n3 * 4 + 2 * wc + jp) * 128 + lq * 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), rsrc_y, (int)(prow[i] + col), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), rsrc_y, (int)(prow[i] + col + 64u), 0, 0);
      }
    if constexpr (BWD) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cs = sum_lr(c3v[jp][j]);
          if (lr == 0) atomicAdd(csl + n3 * 128 + 64 * wc + 32 * jp + lq * 8 + j, cs);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  WSTAMP(5);
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(6);
  if (g_bw3_stamps && (threadIdx.x & 63) == 0 && (uw == 0 || uw == 5)) {      // phase-2 cycle split of waves 0 and 5 (shader clock)
    unsigned long long* o = g_bw3_stamps + (size_t)blockIdx.x * 16 + (uw == 0 ? 8 : 12);
    o[0] = tk_wait; o[1] = tk_lds; o[2] = tk_bar; o[3] = tk_work;
  }
#endif
  if constexpr (BWD) {                            // column sums: one global atomic per channel and workgroup
    __syncthreads();
    atomicAdd(p.cs3 + t, csl[t]);
    if (t < P) atomicAdd(p.cs1 + t, csl[4 * P + t]);
    else if (t < 2 * P) atomicAdd(p.cs2 + t - P, csl[4 * P + t]);
  }
}

}  // namespace

#ifdef AOD_TILE_TIMING
extern "C" int aod_dbg_set_bw3_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bw3_stamps), &buf, sizeof(buf)); }
#endif

static int launch_bw3(bool bwd, Bw3Args& a, int B, int H, int W, aod_stream_t stream) {
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck128x3: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * XC * 2 < 0xe0000000ll, "bottleneck128x3: operand larger than 3.5 GiB");
  AOD_CHECK_ARG((const void*)a.x != (const void*)a.y, "bottleneck128x3: the output must not alias the input (the residual is read after neighbouring tiles have stored)");
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck128x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck128x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  if (bwd) hipLaunchKernelGGL(bottleneck128x3_kernel<true>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bottleneck128x3_kernel<false>, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_bottleneck128x3_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                       const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                       void* t2, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck128x3: null pointer");
  Bw3Args a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2;
  return launch_bw3(false, a, B, H, W, stream);
}

extern "C" int aod_bottleneck128x3_bwd(const void* g, int B, int H, int W, const void* wd3, const void* wd2, const void* wd1, const void* act_t2,
                                       const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2,
                                       float* colsum_t1, float* colsum_x, aod_stream_t stream) {
  AOD_CHECK_ARG(g && wd3 && wd2 && wd1 && act_t2 && act_t1 && act_x && gx && gt2 && gt1 && colsum_t2 && colsum_t1 && colsum_x,
                "bottleneck128x3_bwd: null pointer");
  Bw3Args a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)g; a.w1 = (const bf16_t*)wd3; a.w2 = (const bf16_t*)wd2; a.w3 = (const bf16_t*)wd1;
  a.y = (bf16_t*)gx; a.t1 = (bf16_t*)gt2; a.t2 = (bf16_t*)gt1;
  a.m1 = (const bf16_t*)act_t2; a.m2 = (const bf16_t*)act_t1; a.m3 = (const bf16_t*)act_x;
  a.cs1 = colsum_t2; a.cs2 = colsum_t1; a.cs3 = colsum_x;
  return launch_bw3(true, a, B, H, W, stream);
}
