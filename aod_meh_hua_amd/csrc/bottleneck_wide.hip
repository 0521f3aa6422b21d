// Whole ResNet IDENTITY bottleneck of the 128-plane stage (layer2: 512 -> 128 -> 128 -> 512 channels) in ONE kernel:
//   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + x)                (mmdet/models/backbones/resnet.py:262-301)
// As three launches the block moves 268 MB at 16 x 64 x 64 pixels (x read twice -- conv1 input and residual --, the two 128-channel
// intermediates written and re-read) in 102 us; fused, x and y cross HBM once.  Unlike the 64-plane kernel (bottleneck.hip) the filters do
// not fit the LDS beside the tile (conv2: 288 KB, conv3: 128 KB): they are STREAMED per K-step through LDS-DMA rings, like halo_conv.hip.
//
// One 8-wave workgroup = one 8 x 16 pixel tile of one image (waves 4 x 2 over pixel rows x output channels):
//   phase 1  conv1 on the 10 x 18 HALO of the tile (conv2 needs t1 one pixel around it: 41 % recompute): x and the conv1 filter streamed in
//            64-channel K-steps (2 stages); t1 = relu(bn1(.)) -> LDS as bf16, ZERO outside the image (conv2 pads t1);
//   phase 2  conv2 as 18 K-steps (tap, 64-channel half) whose pixel fragments are gathered from t1 in LDS; 16-KB filter slices through a
//            5-slot ring (4 in flight, counted s_waitcnt vmcnt + one raw s_barrier per step); t2 = relu(bn2(.)) -> LDS;
//   phase 3  conv3 in four chunks of 128 output channels, K = 128 from t2 in LDS, 32-KB filter chunks through a 3-slot ring;
//            epilogue in registers: + residual (requested before the first chunk), ReLU, 16-B stores.
// All products are computed transposed with the paired-block row permutation of bottleneck.hip: a lane holds 8 consecutive channels of
// one pixel.  Optional outputs t1 / t2 (the block's two intermediates, interior pixels only): with them the kernel is the FORWARD of a
// training step as well (the backward pass reads them); without (null) nothing but y leaves the CU.
// Same bf16 rounding points and the same K order per output element as the three-launch block.
#include "common.h"

namespace {

struct BnwArgs {
  const bf16_t* x;       // [B*H*W][512]
  const bf16_t* w1;      // [128][512]
  const bf16_t* w2;      // [128][9][128]  (packed forward form [O][R][S][C])
  const bf16_t* w3;      // [512][128]
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  bf16_t* y;             // [B*H*W][512]
  bf16_t* t1;            // [B*H*W][128] or null
  bf16_t* t2;            // [B*H*W][128] or null
  // backward form (BWD instances): ReLU masks = the block's saved activations, and fp32 column sums of the three results
  const bf16_t* m1;      // [B*H*W][P]    mask of the first product  (forward t2)
  const bf16_t* m2;      // [B*H*W][P]    mask of the second product (forward t1)
  const bf16_t* m3;      // [B*H*W][4P]   mask of the third product  (forward x)
  float* cs1; float* cs2; float* cs3;     // [P] [P] [4P], += (atomics)
  int B, H, W, tiles_y, tiles_x;
};

constexpr int P = 128, CIN = 512;
constexpr int TH = 8, TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * HW_;     // 180 halo pixels
constexpr int HROWS = 192;                                               // padded to 12 row blocks of 16
constexpr int XCH = HROWS * 128;                                         // one 64-channel K-step of the halo (24 576)
constexpr int W1CH = P * 128;                                            // ... and of the conv1 filter (16 384)
constexpr int STG1 = XCH + W1CH;                                         // 40 960
constexpr int OFF_T1 = 2 * STG1;                                         // 81 920: t1 as 2 sub-images [192][128 B]
constexpr int T1SUB = HROWS * 128;
constexpr int OFF_VEC = OFF_T1 + 2 * T1SUB;                              // 131 072: s1 b1 s2 b2 [128] s3 b3 [512] fp32
constexpr int OFF_CS3 = OFF_VEC + (4 * P + 2 * 4 * P) * 4;               // 137 216: column sums (backward form): third result [512], first [128],
constexpr int LDS_BYTES = OFF_CS3 + 6 * P * 4;                           // second [128] fp32 -> 140 288
constexpr int W2SLOT = P * 128, W2R = 5;                                 // phase 2 ring over the phase-1 staging area (81 920 = 5 slots)
constexpr int OFF_T2 = 0, T2SUB = TH * TW * 128;                         // t2 [2][128][128 B] = 32 768 over the dead ring
constexpr int OFF_W3 = 2 * T2SUB, W3SLOT = 2 * P * 128, W3R = 3;         // 32 768 .. 131 072: three 32-KB conv3 filter chunks
static_assert(W2R * W2SLOT <= OFF_T1 && OFF_W3 + W3R * W3SLOT <= OFF_VEC && LDS_BYTES <= 160 * 1024, "LDS map");
constexpr unsigned OOB = 0xf0000000u;

#ifdef AOD_TILE_TIMING
// debug build only (tools/dbg/wide_timing.py): per-workgroup wall-clock stamps (100 MHz) at the phase boundaries
__device__ unsigned long long* g_bnw_stamps = nullptr;
#define WSTAMP(k) do { if (g_bnw_stamps && threadIdx.x == 0) g_bnw_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
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

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// bit j of the result: element j of the 16-B piece (8 bf16) is > 0 -- the ReLU mask of 8 channels in 8 bits
__device__ __forceinline__ unsigned pos_bits(const u32x4_t q) {
  unsigned m = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    m |= (__uint_as_float(q[w] << 16) > 0.f ? 1u : 0u) << (2 * w);
    m |= (__uint_as_float(q[w] & 0xffff0000u) > 0.f ? 1u : 0u) << (2 * w + 1);
  }
  return m;
}
// sum over the 16 lanes that share lq (the lanes of one pixel block that hold the same 8 channels) = one DPP row: four rotate-and-add steps
// on the vector ALU, every lane ends up with the total (__shfl_xor would be four ds_bpermute round trips through the LDS pipe per value)
template <int N> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum_lr(float v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}

// BWD: the same three products are the block's DGRAD chain (csrc/conv.hip dgrad epilogue semantics: result = mask > 0 ? acc (+ res) : 0,
// fp32 column sums of the results): x = the finished gradient G of the block output, w1 / w2 / w3 = the scale-folded dgrad filters of
// conv3 / conv2 / conv1, t1 / t2 / y = the gradients of forward t2 / t1 / x, residual = G (the skip branch).
template <bool BWD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck128_kernel(const BnwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wp = uw >> 1, wc = uw & 1;                    // wave grid: 4 (pixel rows) x 2 (output channels)
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const long long npix = (long long)p.B * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(npix * CIN * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, P * CIN * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, P * 9 * P * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, CIN * P * 2, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(npix * CIN * 2), 0x00020000);
  // the optional outputs through descriptors that are EMPTY when the tensor is not wanted: the stores are always issued (the counted waits
  // below rely on that) and dropped by the range check
  const auto rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1, 0, p.t1 ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t2, 0, p.t2 ? (int)(npix * P * 2) : 0, 0x00020000);

  // LDS-DMA lane roles: one wave-instruction fills 8 rows x 8 slots; lane -> row (lane >> 3) of its group, slot lane & 7, source chunk
  // slot ^ key(row); a wave serves row groups uw + 8 i, for which the keys are the same
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
  // BWD: the column sums of the three results are collected in LDS (ds_add_f32) and leave the workgroup as ONE coalesced global atomic per
  // channel at the end (4-lane atomics straight from the epilogues serialised on their 128-1024 addresses: 635 us instead of 75)
  float* const csl = reinterpret_cast<float*>(smem + OFF_CS3);
  if constexpr (BWD) {
    csl[t] = 0.f;
    if (t < 2 * P) csl[4 * P + t] = 0.f;
  }
  if constexpr (!BWD) {
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
  const auto rsrc_m1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m1, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m2, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m3, 0, BWD ? (int)(npix * CIN * 2) : 0, 0x00020000);
  u32x4_t mk1[3][2];                               // BWD: mask pieces of this lane's halo pixels (out-of-image pixels read 0 = masked)
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int h = (3 * wp + i) * 16 + lr;
      int y, x;
      const unsigned hrow = halo_pix(h, y, x) ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) mk1[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m1, (int)(hrow + (unsigned)((64 * wc + 32 * jp + lq * 8) * 2)), 0, 0);
    }
  }
  unsigned xoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int h = 8 * (uw + 8 * i) + drow;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * CIN + kc * 8) * 2) : OOB;
  }
  unsigned w1off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w1off[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * CIN + kcw * 8) * 2);
  auto issue1 = [&](int buf) {
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
  constexpr int NK1 = CIN / 64;
  issue1(0);
  issue1(1);
  for (int kt = 0; kt < NK1; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < NK1) wait_vm<5>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage kt has landed
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = smem + buf * STG1;
    const char* ws = xs + XCH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 3; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + swz((3 * wp + i) * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading stage kt: its buffer can be refilled
    if (kt + 2 < NK1) issue1(buf);
  }
  WSTAMP(1);
  // epilogue 1: t1 -> LDS (zero outside the image) and, for the tile's own pixels, -> global (training forward)
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
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = 64 * wc + 32 * jp + lq * 8;
        bf16x8 o;
        if constexpr (BWD) {
          const unsigned mb = pos_bits(mk1[i][jp]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ((mb >> (4 * q + r)) & 1u) ? acc1[i][2 * jp + q][r] + 0.f : 0.f;
              o[4 * q + r] = (bf16_t)v;
              cs1v[jp][4 * q + r] += inner ? v : 0.f;
            }
        } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f);
        }
        }
        *reinterpret_cast<bf16x8*>(t1 + wc * T1SUB + swz(h, jp * 4 + lq)) = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t1, (int)(grow + (unsigned)(c * 2)), 0, 0);
      }
    }
    if constexpr (BWD) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = sum_lr(cs1v[jp][j]);
          if (lr == 0) atomicAdd(csl + 4 * P + 64 * wc + 32 * jp + lq * 8 + j, v);
        }
    }
  }
  __syncthreads();                                // t1 complete; the staging area is free
  WSTAMP(2);
  // BWD: the masks of the two later epilogues, requested ahead of the filter ring (older than every ring load: in-order return has them
  // in registers by the first counted wait); the third one (forward x, 16 pieces) is boiled down to a bit per element after the loop
  u32x4_t mk2[2][2], mk3[2][4][2];
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = ty0 + 2 * wp + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      const long long pix = img0 + (long long)y * p.W + x;
      const unsigned r2 = ok ? (unsigned)(pix * (P * 2)) : OOB, r3 = ok ? (unsigned)(pix * (CIN * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        mk2[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m2, (int)(r2 + (unsigned)((64 * wc + 32 * jp + lq * 8) * 2)), 0, 0);
#pragma unroll
        for (int n3 = 0; n3 < 4; ++n3)
          mk3[i][n3][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m3, (int)(r3 + (unsigned)((n3 * 128 + 64 * wc + 32 * jp + lq * 8) * 2)), 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ phase 2: t2 = relu(bn2(conv2(t1))), filter slices through a ring
  unsigned w2base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w2base[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * 9 * P + kcw * 8) * 2);
  auto issue2 = [&](int s, int slot) {           // K-step s = tap * 2 + half
    const unsigned koff = (unsigned)(((s >> 1) * P + (s & 1) * 64) * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(smem + slot * W2SLOT + (uw + 8 * i) * 1024), 16,
                                               w2base[i] + koff, 0, 0, 0);
  };
  f32x4 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NS2 = 18, L2 = W2R - 1;
#pragma unroll
  for (int s = 0; s < L2; ++s) issue2(s, s);
  {
    int slot = 0;
    for (int s = 0; s < NS2; ++s) {
      const int ahead = (NS2 - 1 - s) < (L2 - 1) ? (NS2 - 1 - s) : (L2 - 1);
      wait_vm_dyn(ahead * 2);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + L2 < NS2) issue2(s + L2, slot == 0 ? W2R - 1 : slot - 1);
      const int tap = s >> 1, half = s & 1;
      const int r0_ = tap / 3, q0_ = tap - r0_ * 3;
      const int r = BWD ? 2 - r0_ : r0_, q = BWD ? 2 - q0_ : q0_;      // dgrad: tap (r, s) of the [I][R][S][O] pack reads pixel (y + 1 - r, x + 1 - s)
      const char* t1 = smem + OFF_T1 + half * T1SUB;
      const char* ws = smem + slot * W2SLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[4], af[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(4 * wc + j, lr), ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t1 + swz((2 * wp + i + r) * HW_ + lr + q, ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc2[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = slot == W2R - 1 ? 0 : slot + 1;
    }
  }
  __syncthreads();                                // the ring is dead: t2 goes over it
  WSTAMP(3);
  unsigned mb3[2][2] = {};                        // BWD: byte (n3 & 3) of mb3[i][jp] = mask bits of chunk n3
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int n3 = 0; n3 < 4; ++n3) mb3[i][jp] |= pos_bits(mk3[i][n3][jp]) << (8 * n3);
  }
  unsigned prow[2];                               // byte offset of the lane's pixel rows in a [B*H*W][512] bf16 tensor
  {
    char* t2 = smem + OFF_T2;
    float cs2v[2][8] = {};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = ty0 + 2 * wp + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      prow[i] = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (CIN * 2)) : OOB;
      const unsigned grow = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      const int o_ = (2 * wp + i) * 16 + lr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = 64 * wc + 32 * jp + lq * 8;
        bf16x8 o;
        if constexpr (BWD) {
          const unsigned mb = pos_bits(mk2[i][jp]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ((mb >> (4 * q + r)) & 1u) ? acc2[i][2 * jp + q][r] + 0.f : 0.f;
              o[4 * q + r] = (bf16_t)v;
              cs2v[jp][4 * q + r] += v;
            }
        } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][2 * jp + q][r] * sc[r] + sh[r], 0.f);
        }
        }
        *reinterpret_cast<bf16x8*>(t2 + wc * T2SUB + swz(o_, jp * 4 + lq)) = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t2, (int)(grow + (unsigned)(c * 2)), 0, 0);
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
  __syncthreads();

  WSTAMP(4);
  // ------------------------------------------------------------------ phase 3: y = relu(bn3(conv3(t2)) + x), 4 chunks of 128 output channels
  // residual pieces of the whole tile row pair (16 B each: [pixel row i][chunk][channel pair jp]), requested ahead of the filter chunks
  u32x4_t rv[2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
        rv[i][n3][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(prow[i] + (unsigned)((n3 * 128 + 64 * wc + 32 * jp + lq * 8) * 2)), 0, 0);
  auto issue3 = [&](int n3, int slot) {           // filter chunk n3: rows n3 * 128 .. + 127, two 64-channel sub-images
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = uw + 8 * i, sub = g >> 4, rg = g & 15;
      const unsigned off = (unsigned)((((n3 * 128 + 8 * rg + drow) * P) + sub * 64 + kcw * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(smem + OFF_W3 + slot * W3SLOT + sub * (P * 128) + rg * 1024),
                                               16, off, 0, 0, 0);
    }
  };
  issue3(0, 0);
  issue3(1, 1);
#pragma unroll
  for (int n3 = 0; n3 < 4; ++n3) {
    // younger than this chunk's filter: the next chunk's 4 pieces and the 4 output stores of the previous chunk
    wait_vm_dyn((n3 + 1 < 4 ? 4 : 0) + (n3 > 0 ? 4 : 0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (n3 + 2 < 4) issue3(n3 + 2, (n3 + 2) % W3R);
    f32x4 acc3[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* ws = smem + OFF_W3 + (n3 % W3R) * W3SLOT;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int sub = kb >> 1, ks = kb & 1;
      bf16x8 wf[4], af[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + sub * (P * 128) + wswz(wrow(4 * wc + j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + OFF_T2 + sub * T2SUB + swz((2 * wp + i) * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc3[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float c3v[2][8] = {};                         // BWD: this lane's column sums of the chunk (its two pixel rows)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = n3 * 128 + 64 * wc + 32 * jp + lq * 8;
        const u32x4_t q = rv[i][n3][jp];
        bf16x8 o;
        if constexpr (BWD) {
          const unsigned mb = mb3[i][jp] >> (8 * n3);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc3[i][2 * jp + h2][r] + 0.f;
            v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
            v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = ((mb >> (4 * h2 + r)) & 1u) ? v[r] : 0.f;
              o[4 * h2 + r] = (bf16_t)v[r];
              c3v[jp][4 * h2 + r] += v[r];
            }
          }
        } else
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
    if constexpr (BWD) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cs = sum_lr(c3v[jp][j]);
          if (lr == 0) atomicAdd(csl + n3 * 128 + 64 * wc + 32 * jp + lq * 8 + j, cs);
        }
    }
  }
  WSTAMP(5);
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(6);
#endif
  if constexpr (BWD) {                            // column sums of the third result: one global atomic per channel and workgroup
    __syncthreads();
    atomicAdd(p.cs3 + t, csl[t]);
    if (t < P) atomicAdd(p.cs1 + t, csl[4 * P + t]);
    else if (t < 2 * P) atomicAdd(p.cs2 + t - P, csl[4 * P + t]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 256-plane stage (layer3: 1024 -> 256 -> 256 -> 1024; 16 x 32 x 32 pixels at the bench size, 96 us per block as three launches).
// Tile = 4 x 16 pixels (halo 6 x 18 = 108 pixels in 7 row blocks); the eight waves split the OUTPUT CHANNELS only (wave w owns the paired
// blocks 2w, 2w + 1 = 32 channels of every pixel block), so every filter fragment is read by exactly one wave.  All three filters stream:
// conv1 in 16 K-steps beside the halo chunks (2 stages), conv2 in 36 (tap, 64-channel quarter) slices and conv3 in 16 (256-channel chunk,
// quarter) slices of 32 KB through 3-slot rings.  Per tile 2.2 MB of filter cross L2 -> LDS: that stream, not the 19 us of MFMA work, sets
// the pace (66-73 GB/s per CU, MI355X_MICROARCH.md 'Indexed rows').
namespace w256 {
constexpr int P = 256, CIN = 1024;
constexpr int TH = 4, TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * HW_;     // 108 halo pixels
constexpr int HROWS = 112, XROWS = 128;                                  // 7 row blocks; the x stage is padded to 16 DMA row groups
constexpr int XCH = XROWS * 128, W1CH = P * 128, STG1 = XCH + W1CH;      // 16 384 + 32 768
constexpr int OFF_T1 = 2 * STG1;                                         // 98 304: t1 as 4 sub-images [112][128 B]
constexpr int T1SUB = HROWS * 128;
constexpr int OFF_VEC = OFF_T1 + 4 * T1SUB;                              // 155 648: s1 b1 s2 b2 [256] fp32
constexpr int LDS_BYTES = OFF_VEC + 4 * P * 4;                           // 159 744
constexpr int WSLOT = P * 128, WR = 3;                                   // 32-KB filter slices, 3-slot rings (phase 2 over the staging area)
constexpr int T2SUB = TH * TW * 128;                                     // t2 [4][64][128 B] = 32 768 at offset 0
constexpr int OFF_W3 = 4 * T2SUB;                                        // phase 3 ring 32 768 .. 131 072
constexpr int OFF_VEC3 = OFF_W3 + WR * WSLOT;                            // s3 b3 [1024] fp32: 131 072 .. 139 264 (over the dead t1)
static_assert(WR * WSLOT <= OFF_T1 && OFF_VEC3 + 2 * CIN * 4 <= OFF_VEC && LDS_BYTES <= 160 * 1024, "LDS map");

template <bool BWD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck256_kernel(const BnwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const long long npix = (long long)p.B * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(npix * CIN * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, P * CIN * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, P * 9 * P * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, CIN * P * 2, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(npix * CIN * 2), 0x00020000);
  const auto rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1, 0, p.t1 ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t2, 0, p.t2 ? (int)(npix * P * 2) : 0, 0x00020000);
  const int drow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7);
  const int kcw = kc ^ (((uw >> 1) & 1) << 1);
  auto halo_pix = [&](int h, int& y, int& x) -> bool {
    const int hy = h / HW_, hx = h - hy * HW_;
    y = ty0 - 1 + hy; x = tx0 - 1 + hx;
    return h < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
  };
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if constexpr (!BWD) {
    vec[t] = t < 256 ? p.s1[t] : p.b1[t - 256];
    vec[512 + t] = t < 256 ? p.s2[t] : p.b2[t - 256];
  }
  const float* const vs1 = vec, * const vb1 = vec + 256, * const vs2 = vec + 512, * const vb2 = vec + 768;
  WSTAMP(0);

  // ------------------------------------------------------------------ phase 1
  const int cw = 32 * uw + lq * 8;                 // this lane's 8 consecutive channels (within P, and within a 256-channel conv3 chunk)
  const auto rsrc_m1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m1, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m2, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m3, 0, BWD ? (int)(npix * CIN * 2) : 0, 0x00020000);
  u32x4_t mk1[7];
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      int y, x;
      const unsigned hrow = halo_pix(i * 16 + lr, y, x) ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      mk1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m1, (int)(hrow + (unsigned)(cw * 2)), 0, 0);
    }
  }
  unsigned xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int h = 8 * (uw + 8 * i) + drow;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * CIN + kc * 8) * 2) : OOB;
  }
  unsigned w1off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w1off[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * CIN + kcw * 8) * 2);
  auto issue1 = [&](int buf) {
    char* xs = smem + buf * STG1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = xoff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(xs + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      xoff[i] += 128;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = w1off[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(xs + XCH + (uw + 8 * i) * 1024), 16, off, 0, 0, 0);
      w1off[i] += 128;
    }
  };
  f32x4 acc1[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NK1 = CIN / 64;
  issue1(0);
  issue1(1);
  for (int kt = 0; kt < NK1; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < NK1) wait_vm<6>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = smem + buf * STG1;
    const char* ws = xs + XCH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[2], xf[7];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(2 * uw + j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 7; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + swz(i * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < NK1) issue1(buf);
  }
  WSTAMP(1);
  {
    char* t1 = smem + OFF_T1;
    float cs1v[8] = {};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int h = i * 16 + lr;
      int y, x;
      const bool ok = halo_pix(h, y, x);
      const int hy = h / HW_, hx = h - hy * HW_;
      const bool inner = ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      bf16x8 o;
      if constexpr (BWD) {
        const unsigned mb = pos_bits(mk1[i]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ((mb >> (4 * q + r)) & 1u) ? acc1[i][q][r] + 0.f : 0.f;
            o[4 * q + r] = (bf16_t)v;
            cs1v[4 * q + r] += inner ? v : 0.f;
          }
      } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + cw + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][q][r] * sc[r] + sh[r], 0.f) : 0.f);
      }
      }
      *reinterpret_cast<bf16x8*>(t1 + (uw >> 1) * T1SUB + swz(h, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t1, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {           // (every channel belongs to one wave: plain LDS stores)
        const float v = sum_lr(cs1v[j]);
        if (lr == 0) reinterpret_cast<float*>(smem + OFF_VEC)[cw + j] = v;
      }
    }
  }
  __syncthreads();
  WSTAMP(2);
  u32x4_t mk2[4], mk3[4][4];                       // BWD: masks of the two later epilogues, requested ahead of the filter ring
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      const long long pix = img0 + (long long)y * p.W + x;
      const unsigned r2 = ok ? (unsigned)(pix * (P * 2)) : OOB, r3 = ok ? (unsigned)(pix * (CIN * 2)) : OOB;
      mk2[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m2, (int)(r2 + (unsigned)(cw * 2)), 0, 0);
#pragma unroll
      for (int n3 = 0; n3 < 4; ++n3) mk3[i][n3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m3, (int)(r3 + (unsigned)((n3 * 256 + cw) * 2)), 0, 0);
    }
  }

  // ------------------------------------------------------------------ phase 2: 36 slices (tap, 64-channel quarter)
  unsigned w2base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w2base[i] = (unsigned)(((8 * (uw + 8 * i) + drow) * 9 * P + kcw * 8) * 2);
  auto issue2 = [&](int s, int slot) {           // s = tap * 4 + quarter
    const unsigned koff = (unsigned)(((s >> 2) * P + (s & 3) * 64) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(smem + slot * WSLOT + (uw + 8 * i) * 1024), 16,
                                               w2base[i] + koff, 0, 0, 0);
  };
  f32x4 acc2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NS2 = 36, LA = WR - 1;
  issue2(0, 0);
  issue2(1, 1);
  {
    int slot = 0;
    for (int s = 0; s < NS2; ++s) {
      wait_vm_dyn(s + 1 < NS2 ? 4 : 0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + LA < NS2) issue2(s + LA, slot == 0 ? WR - 1 : slot - 1);
      const int tap = s >> 2, qt = s & 3;
      const int r0_ = tap / 3, q0_ = tap - r0_ * 3;
      const int r = BWD ? 2 - r0_ : r0_, q = BWD ? 2 - q0_ : q0_;
      const char* t1 = smem + OFF_T1 + qt * T1SUB;
      const char* ws = smem + slot * WSLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[2], af[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(2 * uw + j, lr), ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t1 + swz((i + r) * HW_ + lr + q, ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc2[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = slot == WR - 1 ? 0 : slot + 1;
    }
  }
  __syncthreads();
  WSTAMP(3);
  unsigned mb3[4] = {};                            // BWD: byte n3 of mb3[i] = mask bits of chunk n3
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n3 = 0; n3 < 4; ++n3) mb3[i] |= pos_bits(mk3[i][n3]) << (8 * n3);
  }
  unsigned prow[4];
  {
    char* t2 = smem;
    float cs2v[8] = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      prow[i] = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (CIN * 2)) : OOB;
      const unsigned grow = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      bf16x8 o;
      if constexpr (BWD) {
        const unsigned mb = pos_bits(mk2[i]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ((mb >> (4 * q + r)) & 1u) ? acc2[i][q][r] + 0.f : 0.f;
            o[4 * q + r] = (bf16_t)v;
            cs2v[4 * q + r] += v;
          }
      } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + cw + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][q][r] * sc[r] + sh[r], 0.f);
      }
      }
      *reinterpret_cast<bf16x8*>(t2 + (uw >> 1) * T2SUB + swz(i * 16 + lr, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t2, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = sum_lr(cs2v[j]);
        if (lr == 0) reinterpret_cast<float*>(smem + OFF_VEC)[P + cw + j] = v;
      }
    } else {
      // conv3's folded BN vectors over the dead t1 (plain loads: they are waited for right here, before any LDS-DMA of this phase)
      float* const v3 = reinterpret_cast<float*>(smem + OFF_VEC3);
      v3[t] = p.s3[t]; v3[512 + t] = p.s3[512 + t];
      v3[1024 + t] = p.b3[t]; v3[1536 + t] = p.b3[512 + t];
    }
  }
  __syncthreads();

  WSTAMP(4);
  // ------------------------------------------------------------------ phase 3: 4 chunks of 256 output channels x 4 K quarters
  const float* const vs3 = reinterpret_cast<const float*>(smem + OFF_VEC3), * const vb3 = vs3 + 1024;
  u32x4_t rv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) rv[i][n3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(prow[i] + (unsigned)((n3 * 256 + cw) * 2)), 0, 0);
  auto issue3 = [&](int s, int slot) {           // s = chunk * 4 + quarter: rows chunk * 256 .. + 255, K columns quarter * 64 .. + 63
    const int n3 = s >> 2, qt = s & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (unsigned)((((n3 * 256 + 8 * (uw + 8 * i) + drow) * P) + qt * 64 + kcw * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(smem + OFF_W3 + slot * WSLOT + (uw + 8 * i) * 1024), 16,
                                               off, 0, 0, 0);
    }
  };
  constexpr int NS3 = 16;
  issue3(0, 0);
  issue3(1, 1);
  f32x4 acc3[4][2];
  float cs3v[4][8] = {};                           // BWD: column sums of the third result; every channel of it belongs to one wave only
  {
    int slot = 0;
#pragma unroll
    for (int s = 0; s < NS3; ++s) {
      const int n3 = s >> 2, qt = s & 3;
      // younger than this slice: the next slice, and the 4 output stores of a chunk that ended one or two steps ago
      wait_vm_dyn((s + 1 < NS3 ? 4 : 0) + ((s >= 4 && (s & 3) <= 1) ? 4 : 0));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + LA < NS3) issue3(s + LA, slot == 0 ? WR - 1 : slot - 1);
      if (qt == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const char* ws = smem + OFF_W3 + slot * WSLOT;
      const char* t2 = smem + qt * T2SUB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[2], af[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(2 * uw + j, lr), ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t2 + swz(i * 16 + lr, ks * 4 + lq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc3[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (qt == 3) {
        const int c = n3 * 256 + cw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4_t q = rv[i][n3];
          bf16x8 o;
          if constexpr (BWD) {
            const unsigned mb = mb3[i] >> (8 * n3);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc3[i][h2][r] + 0.f;
              v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
              v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = ((mb >> (4 * h2 + r)) & 1u) ? v[r] : 0.f;
                o[4 * h2 + r] = (bf16_t)v[r];
                cs3v[n3][4 * h2 + r] += v[r];
              }
            }
          } else
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc3[i][h2][r] * sc[r] + sh[r];
            v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
            v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * h2 + r] = (bf16_t)fmaxf(v[r], 0.f);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_y, (int)(prow[i] + (unsigned)(c * 2)), 0, 0);
        }
      }
      slot = slot == WR - 1 ? 0 : slot + 1;
    }
  }
  WSTAMP(5);
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(6);
#endif
  if constexpr (BWD) {
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = sum_lr(cs3v[n3][j]);
        if (lr == 0) reinterpret_cast<float*>(smem + OFF_VEC3)[n3 * 256 + cw + j] = v;
      }
    // the three column-sum vectors leave the workgroup as coalesced global atomics, one per channel (4-lane atomics straight from the
    // epilogues serialise on their addresses: 205 us instead of 67).  OFF_VEC / OFF_VEC3: the BN vector areas, unused in this form
    __syncthreads();
    const float* const c12 = reinterpret_cast<const float*>(smem + OFF_VEC);
    const float* const c3 = reinterpret_cast<const float*>(smem + OFF_VEC3);
    if (t < P) atomicAdd(p.cs1 + t, c12[t]); else atomicAdd(p.cs2 + t - P, c12[t]);
    atomicAdd(p.cs3 + t, c3[t]);
    atomicAdd(p.cs3 + 512 + t, c3[512 + t]);
  }
}
}  // namespace w256

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// 256-plane block, second form: FILTERS THROUGH REGISTERS.  Phase stamps of the ring form (tools/dbg/wide_timing.py, profiles/
// r03_wide_phases.txt): every phase ran at the rate of its LDS-DMA filter stream -- 32 KB slices, two in flight = 64 KB per CU against
// ~1.3 us of L2 latency = 49 GB/s -- 23.9 us for conv2 whose MFMA work is 8.8 us.  The ring cannot be deeper (t1 needs the LDS).  But the
// eight waves split the OUTPUT CHANNELS, so a filter fragment is needed by exactly one wave, in exactly the MFMA A-operand layout a plain
// 16-B global load per lane delivers (lane = filter row, 8 consecutive K elements): the filters never touch the LDS.  Each wave prefetches
// its fragments D steps ahead into a register ring (D x 4 KB per wave: 128-192 KB in flight per CU), the K loops of conv2 / conv3 have no
// barrier at all (t1 / t2 are read-only there), conv1 keeps one per step for the x halo chunk, which goes global -> registers -> LDS
// (two stages).  All loads are ordinary loads: the compiler's own s_waitcnt counting applies, no hand-counted vmcnt.
// The filters come as FRAGMENT-MAJOR images (aod_frag_pack): [K-step][wave][j][ks][lane][8 elements], the order the loads consume them.
namespace w256r {
constexpr int P = 256, CIN = 1024;
constexpr int TH = 4, TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * HW_;     // 108 halo pixels
constexpr int HROWS = 112;                                               // 7 row blocks
constexpr int T1SUB = HROWS * 128;                                       // t1 as 4 sub-images [112][128 B]
constexpr int OFF_T1 = 0;
constexpr int XSTG = 128 * 128;                                          // one 64-channel chunk of the halo, 128 rows (112 used)
constexpr int OFF_X = 4 * T1SUB;                                         // 57 344: two x stages
constexpr int T2SUB = TH * TW * 128;                                     // t2 [4][64][128 B] = 32 768 over the x stages
constexpr int OFF_T2 = OFF_X;
constexpr int OFF_VEC = OFF_X + 2 * XSTG;                                // 90 112: s1 b1 s2 b2 [256] s3 b3 [1024] fp32 / column sums (BWD)
constexpr int LDS_BYTES = OFF_VEC + (4 * P + 2 * CIN) * 4;               // 102 400
constexpr int D1 = 3, D2 = 6, D3 = 4, XD = 3;                            // prefetch distances (steps): conv1 / conv2 / conv3 filters, x chunks
static_assert(4 * T2SUB <= 2 * XSTG, "LDS map");

__device__ __forceinline__ bf16x8 as_frag(const u32x4_t v) { return __builtin_bit_cast(bf16x8, v); }

template <bool BWD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck256r_kernel(const BnwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
  const long long img0 = (long long)b * p.H * p.W;
  const long long npix = (long long)p.B * p.H * p.W;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(npix * CIN * 2), 0x00020000);
  const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, P * CIN * 2, 0x00020000);
  const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, P * 9 * P * 2, 0x00020000);
  const auto rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, CIN * P * 2, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(npix * CIN * 2), 0x00020000);
  const auto rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1, 0, p.t1 ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t2, 0, p.t2 ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m1, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m2, 0, BWD ? (int)(npix * P * 2) : 0, 0x00020000);
  const auto rsrc_m3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.m3, 0, BWD ? (int)(npix * CIN * 2) : 0, 0x00020000);
  auto halo_pix = [&](int h, int& y, int& x) -> bool {
    const int hy = h / HW_, hx = h - hy * HW_;
    y = ty0 - 1 + hy; x = tx0 - 1 + hx;
    return h < HPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
  };
  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if constexpr (!BWD) {
    vec[t] = t < 256 ? p.s1[t] : p.b1[t - 256];
    vec[512 + t] = t < 256 ? p.s2[t] : p.b2[t - 256];
    vec[1024 + t] = p.s3[t]; vec[1536 + t] = p.s3[512 + t];
    vec[2048 + t] = p.b3[t]; vec[2560 + t] = p.b3[512 + t];
  }
  const float* const vs1 = vec, * const vb1 = vec + 256, * const vs2 = vec + 512, * const vb2 = vec + 768, * const vs3 = vec + 1024, * const vb3 = vec + 2048;
  WSTAMP(0);
  const int cw = 32 * uw + lq * 8;                 // this lane's 8 consecutive channels (within P, and within a 256-channel conv3 chunk)
  const unsigned wlane = (unsigned)(uw * 4096 + lane * 16);             // this lane's 16 B inside a step's 32 KB of the fragment-major image
  // K-step s of a filter: this wave's four fragments [j][ks] = four 1-KB runs of the fragment-major image (frag_pack_kernel): every load
  // instruction reads 8 whole cache lines.  (Reading the fragments from the row-major pack -- 16 rows x 64 B per instruction -- ran at 32 GB/s
  // per CU, slower than the LDS ring: the L1 handles a line per ~4 clocks whatever part of it is used.)  The step offset travels in the
  // scalar offset operand: a per-step vector address would cost a VGPR per load in flight.
  auto ldw = [&](const auto& rsrc, int s, u32x4_t (&d)[4]) {
#pragma unroll
    for (int f = 0; f < 4; ++f) d[f] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(wlane + f * 1024), s * 32768, 0);
  };

  // ------------------------------------------------------------------ phase 1: conv1 on the halo; x chunks global -> registers -> LDS
  // BWD: the ReLU-mask pieces of an epilogue are requested a K-step or two before it and waited for with a full s_waitcnt right there
  // (like the residual pieces of phase 3: short register lifetimes; registers parked across a phase came back corrupted, see below)
  u32x4_t mk1[7];
  unsigned xoff[2];
  int xlds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int h = 8 * (uw + 8 * i) + (lane >> 3), c = lane & 7;
    int y, x;
    xoff[i] = halo_pix(h, y, x) ? (unsigned)(((img0 + (long long)y * p.W + x) * CIN + c * 8) * 2) : OOB;
    xlds[i] = swz(h, c);
  }
  constexpr int NK1 = CIN / 64;
  u32x4_t xq[XD][2], wq1[D1][4];
#pragma unroll
  for (int k = 0; k < XD; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) xq[k][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)xoff[i], k * 128, 0);
#pragma unroll
  for (int k = 0; k < D1; ++k) ldw(rsrc_w1, k, wq1[k]);
  // chunk 0 -> stage 0
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(smem + OFF_X + xlds[i]) = xq[0][i];
  f32x4 acc1[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NK1; ++kt) {
    // this wave's LDS reads of chunk kt - 1 and its writes of chunk kt are complete; after the barrier every wave's are
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* xs = smem + OFF_X + (kt & 1) * XSTG;
    if (kt + XD < NK1) {        // the register slot of chunk kt was stored to the LDS an iteration ago: refill it
#pragma unroll
      for (int i = 0; i < 2; ++i) xq[kt % XD][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)xoff[i], (kt + XD) * 128, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + swz(i * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wq1[kt % D1][2 * j + ks]), xf[i], acc1[i][j], 0, 0, 0);
    }
    if (kt + D1 < NK1) ldw(rsrc_w1, kt + D1, wq1[kt % D1]);
    if (BWD && kt == NK1 - 2) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        int y, x;
        const unsigned hrow = halo_pix(i * 16 + lr, y, x) ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
        mk1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m1, (int)(hrow + (unsigned)(cw * 2)), 0, 0);
      }
    }
    if (kt + 1 < NK1) {         // chunk kt + 1 -> the other stage (its last readers passed this iteration's barrier)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(smem + OFF_X + ((kt + 1) & 1) * XSTG + xlds[i]) = xq[(kt + 1) % XD][i];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  WSTAMP(1);
  if constexpr (BWD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the mask pieces of epilogue 1
  // conv2's first filter steps are requested before the epilogue
  u32x4_t wq2[D2][4];
#pragma unroll
  for (int k = 0; k < D2; ++k) ldw(rsrc_w2, k, wq2[k]);
  {
    char* t1 = smem + OFF_T1;
    float cs1v[8] = {};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int h = i * 16 + lr;
      int y, x;
      const bool ok = halo_pix(h, y, x);
      const int hy = h / HW_, hx = h - hy * HW_;
      const bool inner = ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      bf16x8 o;
      if constexpr (BWD) {
        const unsigned mb = pos_bits(mk1[i]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ((mb >> (4 * q + r)) & 1u) ? acc1[i][q][r] + 0.f : 0.f;
            o[4 * q + r] = (bf16_t)v;
            cs1v[4 * q + r] += inner ? v : 0.f;
          }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + cw + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][q][r] * sc[r] + sh[r], 0.f) : 0.f);
        }
      }
      *reinterpret_cast<bf16x8*>(t1 + (uw >> 1) * T1SUB + swz(h, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t1, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = sum_lr(cs1v[j]);
        if (lr == 0) vec[cw + j] = v;
      }
    }
  }
  u32x4_t mk2[4], mk3[4][4];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // t1 complete
  asm volatile("" ::: "memory");
  WSTAMP(2);

  // ------------------------------------------------------------------ phase 2: conv2, 36 (tap, 64-channel quarter) steps, no barrier
  f32x4 acc2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NS2 = 36;
#pragma unroll
  for (int s = 0; s < NS2; ++s) {
    const int tap = s >> 2, qt = s & 3;
    const int r0_ = tap / 3, q0_ = tap - r0_ * 3;
    const int r = BWD ? 2 - r0_ : r0_, q = BWD ? 2 - q0_ : q0_;
    const char* t1 = smem + OFF_T1 + qt * T1SUB;
    // LOADS return in order, but stores complete in any order relative to them, and hipcc's own vmcnt(N) counts the t1 stores of the
    // epilogue (younger than the first D2 steps' loads) as if they were in order too: N may be reached by stores finishing early.  A count
    // of the younger LOADS only is safe whatever the stores do (here: the D2 - 1 later steps that are in flight).
    wait_vm_dyn(4 * ((NS2 - 1 - s) < (D2 - 1) ? (NS2 - 1 - s) : (D2 - 1)));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t1 + swz((i + r) * HW_ + lr + q, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wq2[s % D2][2 * j + ks]), af[i], acc2[i][j], 0, 0, 0);
    }
    if (s + D2 < NS2) ldw(rsrc_w2, s + D2, wq2[s % D2]);
    if (BWD && s == NS2 - 2) {      // the mask pieces of the two later epilogues
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = ty0 + i, x = tx0 + lr;
        const bool ok = y < p.H && x < p.W;
        const long long pix = img0 + (long long)y * p.W + x;
        const unsigned r2 = ok ? (unsigned)(pix * (P * 2)) : OOB, r3 = ok ? (unsigned)(pix * (CIN * 2)) : OOB;
        mk2[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m2, (int)(r2 + (unsigned)(cw * 2)), 0, 0);
#pragma unroll
        for (int n3 = 0; n3 < 4; ++n3) mk3[i][n3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_m3, (int)(r3 + (unsigned)(cw * 2)), n3 * 512, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  WSTAMP(3);
  unsigned mb3[4] = {};                            // BWD: byte n3 of mb3[i] = mask bits of chunk n3 (64 registers of mask pieces -> 4)
  if constexpr (BWD) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n3 = 0; n3 < 4; ++n3) mb3[i] |= pos_bits(mk3[i][n3]) << (8 * n3);
  }
  // conv3's first filter steps and the residual pieces are requested before the epilogue
  u32x4_t wq3[D3][4];
#pragma unroll
  for (int k = 0; k < D3; ++k) ldw(rsrc_w3, k, wq3[k]);
  unsigned prow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = ty0 + i, x = tx0 + lr;
    prow[i] = (y < p.H && x < p.W) ? (unsigned)((img0 + (long long)y * p.W + x) * (CIN * 2)) : OOB;
  }
  // residual pieces: requested per 256-channel chunk, one chunk (four K-steps) ahead of their use.  (All 16 pieces up front -- 64 registers
  // parked across the whole phase -- came back corrupted in the second wave of every SIMD on this toolchain: the kernel then sits at the
  // 256-register cap and the allocator parks them in accumulator registers; tools/dbg/frag_check.py.)
  u32x4_t rv[4];
  {
    char* t2 = smem + OFF_T2;
    float cs2v[8] = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned grow = prow[i] == OOB ? OOB : prow[i] / 4;        // row offset in a [.][256] tensor
      bf16x8 o;
      if constexpr (BWD) {
        const unsigned mb = pos_bits(mk2[i]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ((mb >> (4 * q + r)) & 1u) ? acc2[i][q][r] + 0.f : 0.f;
            o[4 * q + r] = (bf16_t)v;
            cs2v[4 * q + r] += v;
          }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + cw + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][q][r] * sc[r] + sh[r], 0.f);
        }
      }
      *reinterpret_cast<bf16x8*>(t2 + (uw >> 1) * T2SUB + swz(i * 16 + lr, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t2, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = sum_lr(cs2v[j]);
        if (lr == 0) vec[P + cw + j] = v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // t2 complete
  asm volatile("" ::: "memory");
  WSTAMP(4);

  // ------------------------------------------------------------------ phase 3: conv3, 4 chunks of 256 output channels x 4 K quarters, no barrier
  constexpr int NS3 = 16;
  f32x4 acc3[4][2];
  float cs3v[4][8] = {};
#pragma unroll
  for (int s = 0; s < NS3; ++s) {
    const int n3 = s >> 2, qt = s & 3;
    if (qt == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const char* t2 = smem + OFF_T2 + qt * T2SUB;
    wait_vm_dyn(4 * ((NS3 - 1 - s) < (D3 - 1) ? (NS3 - 1 - s) : (D3 - 1)));      // (younger loads only: the t2 / y stores complete in any order)
    if (qt == 3) {              // this chunk's residual pieces: requested one K-step before the epilogue that adds them
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(prow[i] + (unsigned)(cw * 2)), n3 * 512, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(t2 + swz(i * 16 + lr, ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wq3[s % D3][2 * j + ks]), af[i], acc3[i][j], 0, 0, 0);
    }
    if (s + D3 < NS3) ldw(rsrc_w3, s + D3, wq3[s % D3]);
    if (qt == 3) {
      const int c = n3 * 256 + cw;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the residual pieces (and, with them, everything older)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4_t q = rv[i];
        bf16x8 o;
        if constexpr (BWD) {
          const unsigned mb = mb3[i] >> (8 * n3);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc3[i][h2][r] + 0.f;
            v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
            v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = ((mb >> (4 * h2 + r)) & 1u) ? v[r] : 0.f;
              o[4 * h2 + r] = (bf16_t)v[r];
              cs3v[n3][4 * h2 + r] += v[r];
            }
          }
        } else {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vs3 + c + 4 * h2), sh = *reinterpret_cast<const f32x4*>(vb3 + c + 4 * h2);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc3[i][h2][r] * sc[r] + sh[r];
            v[0] += __uint_as_float(q[2 * h2] << 16); v[1] += __uint_as_float(q[2 * h2] & 0xffff0000u);
            v[2] += __uint_as_float(q[2 * h2 + 1] << 16); v[3] += __uint_as_float(q[2 * h2 + 1] & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * h2 + r] = (bf16_t)fmaxf(v[r], 0.f);
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_y, (int)(prow[i] + (unsigned)(cw * 2)), n3 * 512, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  WSTAMP(5);
#ifdef AOD_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(6);
#endif
  if constexpr (BWD) {
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = sum_lr(cs3v[n3][j]);
        if (lr == 0) vec[2 * P + n3 * 256 + cw + j] = v;
      }
    __syncthreads();
    if (t < P) atomicAdd(p.cs1 + t, vec[t]); else atomicAdd(p.cs2 + t - P, vec[t]);
    atomicAdd(p.cs3 + t, vec[2 * P + t]);
    atomicAdd(p.cs3 + 512 + t, vec[2 * P + 512 + t]);
  }
}
}  // namespace w256r

// Fragment-major filter images for the register-streamed kernels.  Source: a row-major packed filter [rows][K] bf16 (forward pack
// [O][R][S][C] or dgrad pack [I][R][S][O]; rows % 256 == 0, K % 64 == 0).  With KS = K / 64 steps per block of 256 rows, step s = rb * KS + kq:
//   dst[s][w][j][ks][lane][0..8) = src[256 rb + 32 w + 8 ((lane & 15) >> 2) + 4 j + (lane & 3)][64 kq + 8 (4 ks + (lane >> 4)) + 0..8)
// (the paired-block row permutation of the MFMA A operand, see bottleneck.hip).  One launch packs every listed filter.
struct FragItem { const bf16_t* src; bf16_t* dst; int rows, K, blk0, pad_; };
__global__ __launch_bounds__(256) void frag_pack_kernel(const FragItem* __restrict__ items, int nitems) {
  int lo = 0, hi = nitems - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FragItem it = items[lo];
  const int u = ((int)blockIdx.x - it.blk0) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // 1-KB unit of the image
  const int KS = it.K >> 6;
  if (u >= (it.rows >> 8) * KS * 32) return;
  const int ks = u & 1, j = (u >> 1) & 1, w = (u >> 2) & 7, s = u >> 5;
  const int rb = s / KS, kq = s - rb * KS;
  const int row = 256 * rb + 32 * w + 8 * ((lane & 15) >> 2) + 4 * j + (lane & 3);
  const int col = 64 * kq + 8 * (4 * ks + (lane >> 4));
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(it.src + (long long)row * it.K + col);
  *reinterpret_cast<u32x4_t*>(it.dst + (long long)u * 512 + lane * 8) = v;
}
extern "C" int aod_frag_pack_item_bytes(void) { return (int)sizeof(FragItem); }
extern "C" int aod_frag_pack(const void* items_dev, int nitems, int total_blocks, aod_stream_t stream) {
  if (nitems == 0) return 0;
  AOD_CHECK_ARG(items_dev && nitems > 0 && total_blocks > 0, "frag_pack: bad args");
  hipLaunchKernelGGL(frag_pack_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const FragItem*)items_dev, nitems);
  AOD_LAUNCH_CHECK();
  return 0;
}

#ifdef AOD_TILE_TIMING
extern "C" int aod_dbg_set_bnw_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bnw_stamps), &buf, sizeof(buf)); }
#endif

template <int PL, bool BWD>
static int launch_wide(BnwArgs& a, hipStream_t st, bool frag = false) {
  constexpr int th = PL == 128 ? TH : w256::TH, tw = PL == 128 ? TW : w256::TW, lds = PL == 128 ? LDS_BYTES : w256::LDS_BYTES;
  a.tiles_y = (a.H + th - 1) / th; a.tiles_x = (a.W + tw - 1) / tw;
  const void* fn;
  if constexpr (PL == 128) fn = reinterpret_cast<const void*>(&bottleneck128_kernel<BWD>);
  else fn = reinterpret_cast<const void*>(&w256::bottleneck256_kernel<BWD>);
  static unsigned long long attr_done = 0;               // one flag per instantiation
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (PL == 256)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w256r::bottleneck256r_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, w256r::LDS_BYTES);
  }
  if constexpr (PL == 128) hipLaunchKernelGGL(bottleneck128_kernel<BWD>, dim3(a.B * a.tiles_y * a.tiles_x), dim3(512), lds, st, a);
  else if (!frag) hipLaunchKernelGGL(w256::bottleneck256_kernel<BWD>, dim3(a.B * a.tiles_y * a.tiles_x), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(w256r::bottleneck256r_kernel<BWD>, dim3(a.B * a.tiles_y * a.tiles_x), dim3(512), w256r::LDS_BYTES, st, a);
  AOD_LAUNCH_CHECK();
  return 0;
}

static int wide_fwd(int PL, const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                    const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1, void* t2, aod_stream_t stream,
                    bool frag = false) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck128/256: null pointer");
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck128/256: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * (4 * PL) * 2 < 0xe0000000ll, "bottleneck128/256: operand larger than 3.5 GiB");
  BnwArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2;
  a.B = B; a.H = H; a.W = W;
  return PL == 128 ? launch_wide<128, false>(a, (hipStream_t)stream) : launch_wide<256, false>(a, (hipStream_t)stream, frag);
}

extern "C" int aod_bottleneck128_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                     const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                     void* t2, aod_stream_t stream) {
  return wide_fwd(128, x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, y, t1, t2, stream);
}

extern "C" int aod_bottleneck256_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                     const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                     void* t2, aod_stream_t stream) {
  return wide_fwd(256, x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, y, t1, t2, stream);
}

static int wide_bwd(int planes, bool frag, const void* g, int B, int H, int W, const void* wd3, const void* wd2, const void* wd1,
                    const void* act_t2, const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2,
                    float* colsum_t1, float* colsum_x, aod_stream_t stream) {
  AOD_CHECK_ARG(planes == 128 || planes == 256, "bottleneck_bwd: planes must be 128 or 256");
  AOD_CHECK_ARG(g && wd3 && wd2 && wd1 && act_t2 && act_t1 && act_x && gx && gt2 && gt1 && colsum_t2 && colsum_t1 && colsum_x,
                "bottleneck_bwd: null pointer");
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck_bwd: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * (4 * planes) * 2 < 0xe0000000ll, "bottleneck_bwd: operand larger than 3.5 GiB");
  BnwArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)g; a.w1 = (const bf16_t*)wd3; a.w2 = (const bf16_t*)wd2; a.w3 = (const bf16_t*)wd1;
  a.y = (bf16_t*)gx; a.t1 = (bf16_t*)gt2; a.t2 = (bf16_t*)gt1;
  a.m1 = (const bf16_t*)act_t2; a.m2 = (const bf16_t*)act_t1; a.m3 = (const bf16_t*)act_x;
  a.cs1 = colsum_t2; a.cs2 = colsum_t1; a.cs3 = colsum_x;
  a.B = B; a.H = H; a.W = W;
  return planes == 128 ? launch_wide<128, true>(a, (hipStream_t)stream) : launch_wide<256, true>(a, (hipStream_t)stream, frag);
}

extern "C" int aod_bottleneck_bwd(int planes, const void* g, int B, int H, int W, const void* wd3, const void* wd2, const void* wd1,
                                  const void* act_t2, const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1, float* colsum_t2,
                                  float* colsum_t1, float* colsum_x, aod_stream_t stream) {
  return wide_bwd(planes, false, g, B, H, W, wd3, wd2, wd1, act_t2, act_t1, act_x, gx, gt2, gt1, colsum_t2, colsum_t1, colsum_x, stream);
}

// The 256-plane block with FRAGMENT-MAJOR filter images (aod_frag_pack of the same packs): filters stream through registers, not the LDS.
extern "C" int aod_bottleneck256f_fwd(const void* x, int B, int H, int W, const void* w1f, const float* s1, const float* b1, const void* w2f,
                                      const float* s2, const float* b2, const void* w3f, const float* s3, const float* b3, void* y, void* t1,
                                      void* t2, aod_stream_t stream) {
  return wide_fwd(256, x, B, H, W, w1f, s1, b1, w2f, s2, b2, w3f, s3, b3, y, t1, t2, stream, true);
}
extern "C" int aod_bottleneck256f_bwd(const void* g, int B, int H, int W, const void* wd3f, const void* wd2f, const void* wd1f,
                                      const void* act_t2, const void* act_t1, const void* act_x, void* gx, void* gt2, void* gt1,
                                      float* colsum_t2, float* colsum_t1, float* colsum_x, aod_stream_t stream) {
  return wide_bwd(256, true, g, B, H, W, wd3f, wd2f, wd1f, act_t2, act_t1, act_x, gx, gt2, gt1, colsum_t2, colsum_t1, colsum_x, stream);
}
