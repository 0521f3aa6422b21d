// Frozen ResNet stem in one kernel: max_pool_3x3_s2(relu(bn1(conv1_7x7_s2(image))))  (mmdet/models/backbones/resnet.py:630-637) on the
// space-to-depth image (aod_nchw_f32_to_s2d_bf16: the 7x7 / stride-2 conv is a 4x4 / stride-1 / pad-2 conv over [B][H2][W2][16]).
// As separate launches the 64-channel conv output (134 MB at 16 x 512 x 512) is written and read back by the pooling pass; here it lives
// in LDS.  One 8-wave workgroup = one 8 x 16 tile of POOLED pixels of one image:
//   1  the 20 x 36-pixel input patch (23 KB) and the 64 x 256 filter (32 KB) arrive by LDS-DMA;
//   2  the 17 x 33 conv outputs the pool windows of the tile need (9 % recompute) are computed as 36 row blocks of 16 pixels: the im2col
//      row of a pixel and filter row R is 4 pixels x 32 B = 128 contiguous bytes of the patch, read as fragments straight from it;
//      accumulators transposed with the paired-block channel permutation of bottleneck.hip -> BN + ReLU -> bf16 conv tile in LDS (72 KB),
//      zero outside the image (all values are >= 0 after the ReLU, so a zero pad equals the pool's -inf pad);
//   3  3 x 3 / stride-2 max over the LDS tile, 16-B stores of the pooled pixels.
#include "common.h"

namespace {

struct StemArgs {
  const bf16_t* x;      // [B][H2][W2][16] space-to-depth image
  const bf16_t* w;      // [64][4][4][16] packed forward filter (K = 256)
  const float* scale; const float* shift;   // folded bn1 [64]
  bf16_t* y;            // [B][H4][W4][64] pooled output
  int B, H2, W2, H4, W4, tiles_y, tiles_x;
};

constexpr int PTH = 8, PTW = 16;                    // pooled tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1; // conv outputs behind it: 17 x 33
constexpr int CPIX = CTH * CTW;                     // 561
constexpr int CBLK = (CPIX + 15) / 16;              // 36 row blocks
constexpr int PH = CTH + 3, PW = CTW + 3;           // input patch: 20 x 36 pixels of 32 B
constexpr int OFF_PATCH = 0;
constexpr int PATCH_BYTES = PH * PW * 32;           // 23040
constexpr int OFF_W = 23552;                        // [4 filter rows][64][128 B]
constexpr int OFF_CONV = OFF_W + 4 * 8192;          // [576][128 B]
constexpr int OFF_VEC = OFF_CONV + CBLK * 16 * 128; // scale, shift
constexpr int LDS_BYTES = OFF_VEC + 128 * 4;        // 130560

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// filter images: the paired-block row permutation (wrow) makes the 16 lanes of a fragment read rows {0-3, 8-11, 16-19, 24-27} (+4): rows r and
// r + 16 share (r >> 1) & 7 and lie 2048 B apart -- the same banks.  One more row bit in the XOR key separates them (PMC before: 35 % of
// the LDS cycles of the kernel were bank conflicts).
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_pool_kernel(const StemArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, p.B * ntile);
  const int b = wg / ntile, tt = wg - b * ntile;
  const int py0 = (tt / p.tiles_x) * PTH, px0 = (tt % p.tiles_x) * PTW;   // first pooled pixel of the tile
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;                          // first conv output behind it
  const int iy0 = cy0 - 2, ix0 = cx0 - 2;                                  // first input (space-to-depth) pixel of the patch
  constexpr unsigned OOB = 0xf0000000u;
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.B * p.H2 * p.W2 * 32), 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 64 * 256 * 2, 0x00020000);

  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if (t < 64) vec[t] = p.scale[t];
  else if (t < 128) vec[t] = p.shift[t - 64];

  // ---- 1. patch and filter by LDS-DMA.  The patch image is linear: 16-B unit q = 2 * (row * PW + col) + half
  for (int k = uw; k < (PATCH_BYTES + 1023) / 1024; k += 8) {
    const int q = k * 64 + lane;
    const int px = q >> 1, pr = px / PW, pc = px - pr * PW;
    const int y = iy0 + pr, x = ix0 + pc;
    const bool ok = q < PATCH_BYTES / 16 && (unsigned)y < (unsigned)p.H2 && (unsigned)x < (unsigned)p.W2;
    const unsigned off = ok ? (unsigned)((((long long)b * p.H2 + y) * p.W2 + x) * 32 + (q & 1) * 16) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(smem + OFF_PATCH + k * 1024), 16, off, 0, 0, 0);
  }
  {
    const int drow = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7) ^ (((uw >> 1) & 1) << 1);      // wswz key of rows 8 uw + drow
#pragma unroll
    for (int R = 0; R < 4; ++R) {
      const unsigned off = (unsigned)(((8 * uw + drow) * 256 + R * 64 + kc * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + OFF_W + R * 8192 + uw * 1024), 16, off, 0, 0, 0);
    }
  }
  __syncthreads();

  // ---- 2. conv on the 17 x 33 tile: row blocks rb = uw + 8 i
  f32x4 acc[5][4];
  int abase[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pidx = (uw + 8 * i) * 16 + lr;
    pidx = pidx < CPIX ? pidx : CPIX - 1;                     // (pad rows of the last block: computed, never used)
    const int oy = pidx / CTW, ox = pidx - oy * CTW;
    abase[i] = OFF_PATCH + (oy * PW + ox) * 32 + lq * 16;
  }
#pragma unroll
  for (int R = 0; R < 4; ++R) {
    const char* ws = smem + OFF_W + R * 8192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(ws + wswz(wrow(j, lr), ks * 4 + lq));
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (uw + 8 * i < CBLK) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + abase[i] + R * (PW * 32) + ks * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  {
    char* ct = smem + OFF_CONV;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (uw + 8 * i < CBLK) {
        const int pidx = (uw + 8 * i) * 16 + lr;
        const int oy = pidx / CTW, ox = pidx - oy * CTW;
        const bool ok = pidx < CPIX && (unsigned)(cy0 + oy) < (unsigned)p.H2 && (unsigned)(cx0 + ox) < (unsigned)p.W2;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int c = jp * 32 + lq * 8;
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vec + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vec + 64 + c + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * q + r] = (bf16_t)(ok ? fmaxf(acc[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f);
          }
          *reinterpret_cast<bf16x8*>(ct + swz(pidx, jp * 4 + lq)) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- 3. 3 x 3 / stride-2 max over the conv tile; item = (pooled pixel, 8-channel chunk)
  const char* ct = smem + OFF_CONV;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * 512 + t;
    const int c8 = item & 7, pp = item >> 3, ppy = pp / PTW, ppx = pp - ppy * PTW;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = 0.f;                    // (values are >= 0 after the ReLU)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int row = (2 * ppy + dy) * CTW + 2 * ppx + dx;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ct + swz(row, c8));
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], (float)v[k]);
      }
    const int py = py0 + ppy, px = px0 + ppx;
    if (py < p.H4 && px < p.W4) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (bf16_t)m[k];
      *reinterpret_cast<bf16x8*>(p.y + (((long long)b * p.H4 + py) * p.W4 + px) * 64 + c8 * 8) = o;
    }
  }
}

}  // namespace

extern "C" int aod_stem_pool_fwd(const void* x_s2d, const void* w_packed, const float* scale, const float* shift, void* y, int B, int H2, int W2,
                                 aod_stream_t stream) {
  AOD_CHECK_ARG(x_s2d && w_packed && scale && shift && y && B >= 1 && H2 >= 1 && W2 >= 1, "stem_pool: bad args");
  AOD_CHECK_ARG((long long)B * H2 * W2 * 32 < 0xe0000000ll, "stem_pool: image batch larger than 3.5 GiB");
  StemArgs a;
  a.x = (const bf16_t*)x_s2d; a.w = (const bf16_t*)w_packed; a.scale = scale; a.shift = shift; a.y = (bf16_t*)y;
  a.B = B; a.H2 = H2; a.W2 = W2;
  a.H4 = (H2 - 1) / 2 + 1; a.W4 = (W2 - 1) / 2 + 1;          // max_pool2d(kernel 3, stride 2, padding 1)
  a.tiles_y = (a.H4 + PTH - 1) / PTH; a.tiles_x = (a.W4 + PTW - 1) / PTW;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  hipLaunchKernelGGL(stem_pool_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}
