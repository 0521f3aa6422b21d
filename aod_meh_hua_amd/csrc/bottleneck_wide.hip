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
constexpr int LDS_BYTES = OFF_VEC + (4 * P + 2 * 4 * P) * 4;             // 137 216
constexpr int W2SLOT = P * 128, W2R = 5;                                 // phase 2 ring over the phase-1 staging area (81 920 = 5 slots)
constexpr int OFF_T2 = 0, T2SUB = TH * TW * 128;                         // t2 [2][128][128 B] = 32 768 over the dead ring
constexpr int OFF_W3 = 2 * T2SUB, W3SLOT = 2 * P * 128, W3R = 3;         // 32 768 .. 131 072: three 32-KB conv3 filter chunks
static_assert(W2R * W2SLOT <= OFF_T1 && OFF_W3 + W3R * W3SLOT <= OFF_VEC && LDS_BYTES <= 160 * 1024, "LDS map");
constexpr unsigned OOB = 0xf0000000u;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define AOD_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_dyn(int n) {      // n is wave-uniform
  switch (n) {
    AOD_VMCASE(0) AOD_VMCASE(1) AOD_VMCASE(2) AOD_VMCASE(3) AOD_VMCASE(4) AOD_VMCASE(5) AOD_VMCASE(6) AOD_VMCASE(7) AOD_VMCASE(8)
    AOD_VMCASE(9) AOD_VMCASE(10) AOD_VMCASE(11) AOD_VMCASE(12)
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

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck128_fwd_kernel(const BnwArgs p) {
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
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = 64 * wc + 32 * jp + lq * 8;
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f);
        }
        *reinterpret_cast<bf16x8*>(t1 + wc * T1SUB + swz(h, jp * 4 + lq)) = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t1, (int)(grow + (unsigned)(c * 2)), 0, 0);
      }
    }
  }
  __syncthreads();                                // t1 complete; the staging area is free

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
      const int r = tap / 3, q = tap - r * 3;
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
  unsigned prow[2];                               // byte offset of the lane's pixel rows in a [B*H*W][512] bf16 tensor
  {
    char* t2 = smem + OFF_T2;
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
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + c + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][2 * jp + q][r] * sc[r] + sh[r], 0.f);
        }
        *reinterpret_cast<bf16x8*>(t2 + wc * T2SUB + swz(o_, jp * 4 + lq)) = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t2, (int)(grow + (unsigned)(c * 2)), 0, 0);
      }
    }
  }
  __syncthreads();

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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c = n3 * 128 + 64 * wc + 32 * jp + lq * 8;
        const u32x4_t q = rv[i][n3][jp];
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

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck256_fwd_kernel(const BnwArgs p) {
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
  {
    vec[t] = t < 256 ? p.s1[t] : p.b1[t - 256];
    vec[512 + t] = t < 256 ? p.s2[t] : p.b2[t - 256];
  }
  const float* const vs1 = vec, * const vb1 = vec + 256, * const vs2 = vec + 512, * const vb2 = vec + 768;

  // ------------------------------------------------------------------ phase 1
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
  const int cw = 32 * uw + lq * 8;                 // this lane's 8 consecutive channels (within P, and within a 256-channel conv3 chunk)
  {
    char* t1 = smem + OFF_T1;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int h = i * 16 + lr;
      int y, x;
      const bool ok = halo_pix(h, y, x);
      const int hy = h / HW_, hx = h - hy * HW_;
      const bool inner = ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
      const unsigned grow = inner ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs1 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb1 + cw + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc1[i][q][r] * sc[r] + sh[r], 0.f) : 0.f);
      }
      *reinterpret_cast<bf16x8*>(t1 + (uw >> 1) * T1SUB + swz(h, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t1, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
  }
  __syncthreads();

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
      const int r = tap / 3, q = tap - r * 3;
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
  unsigned prow[4];
  {
    char* t2 = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + i, x = tx0 + lr;
      const bool ok = y < p.H && x < p.W;
      prow[i] = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (CIN * 2)) : OOB;
      const unsigned grow = ok ? (unsigned)((img0 + (long long)y * p.W + x) * (P * 2)) : OOB;
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(vs2 + cw + 4 * q), sh = *reinterpret_cast<const f32x4*>(vb2 + cw + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)fmaxf(acc2[i][q][r] * sc[r] + sh[r], 0.f);
      }
      *reinterpret_cast<bf16x8*>(t2 + (uw >> 1) * T2SUB + swz(i * 16 + lr, (uw & 1) * 4 + lq)) = o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_t2, (int)(grow + (unsigned)(cw * 2)), 0, 0);
    }
    // conv3's folded BN vectors over the dead t1 (plain loads: they are waited for right here, before any LDS-DMA of this phase)
    float* const v3 = reinterpret_cast<float*>(smem + OFF_VEC3);
    v3[t] = p.s3[t]; v3[512 + t] = p.s3[512 + t];
    v3[1024 + t] = p.b3[t]; v3[1536 + t] = p.b3[512 + t];
  }
  __syncthreads();

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
}
}  // namespace w256

}  // namespace

extern "C" int aod_bottleneck128_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                     const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                     void* t2, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck128: null pointer");
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck128: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * CIN * 2 < 0xe0000000ll, "bottleneck128: operand larger than 3.5 GiB");
  BnwArgs a;
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2;
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck128_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(bottleneck128_fwd_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" int aod_bottleneck256_fwd(const void* x, int B, int H, int W, const void* w1, const float* s1, const float* b1, const void* w2,
                                     const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, void* y, void* t1,
                                     void* t2, aod_stream_t stream) {
  AOD_CHECK_ARG(x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && y, "bottleneck256: null pointer");
  AOD_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "bottleneck256: bad geometry");
  AOD_CHECK_ARG((long long)B * H * W * w256::CIN * 2 < 0xe0000000ll, "bottleneck256: operand larger than 3.5 GiB");
  BnwArgs a;
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2;
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + w256::TH - 1) / w256::TH; a.tiles_x = (W + w256::TW - 1) / w256::TW;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w256::bottleneck256_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, w256::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(w256::bottleneck256_fwd_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), w256::LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}
