// Frozen ResNet stem of the REFERENCE-PRECISION mode in one kernel: max_pool_3x3_s2(relu(bn1(conv1_7x7_s2(image))))
// (mmdet/models/backbones/resnet.py:630-637) from the fp32 NCHW image to X-layout rows (x3_ops.hip) of the pooled map.
// As separate launches (aod_x3_nchw_f32_to_s2d -> conv.hip X3 on a 16-tap filter whose 32-channel bands hold 12 real values -> aod_x3_maxpool3x3s2)
// the path writes and re-reads a 134 MB space-to-depth image and a 268 MB conv output at 16 x 512 x 512 and spends 62 % of its matrix work on
// zero padding; here the image is read once, split into heads and tails on the way into LDS, and only the pooled map is written.
// One 8-wave workgroup per CU walks 8 x 16 tiles of POOLED pixels (the tile geometry of stem.hip); per tile:
//   1  the 40 x 72-pixel fp32 image patch is loaded as float2 pairs, re-ordered to space-to-depth slots ((dy, dx, c): the order of
//      aod_nchw_f32_to_s2d_bf16) and written as a heads patch and a tails patch of 20 x 36 pixels x 32 B; the X filter [64][4][4][h 32 | l 32]
//      (12 real slots per tap) is written from registers as a heads and a tails image [4 filter rows][64][4 taps x 16 slots];
//   2  the 17 x 33 conv outputs behind the tile: per filter row the im2col row of a pixel is 128 contiguous bytes of a patch; three MFMAs
//      per K-step (Wh.Ah + Wh.Al + Wl.Ah, fp32 accumulate) -- K = 256 slots per output, 24 MFMAs per 16 x 16 block instead of the 48 of the
//      padded X rows.  The fp32 sums are therefore grouped differently from the three-launch path's: the two agree to fp32 rounding
//      (tests/test_gpu_x3_kernels.py pins 2e-5 of the map's scale: one tail rounding), not bit for bit;
//   3  BN + ReLU, the tile as fp32 in LDS over the patch / filter region (zero outside the image: every value is >= 0); 3 x 3 / stride-2
//      max, then the roundings of the three-launch path in its order (the conv output to head + tail, the pooled value to head + tail:
//      rounding is monotone, so it commutes with the max), head + tail stores of the pooled pixels.
#include "common.h"

namespace {

struct StemX3Args {
  const float* img;     // [B][3][H][W]
  const bf16_t* w;      // X filter [64][4][4][64]
  const float* scale; const float* shift;   // folded bn1 [64]
  bf16_t* y;            // X rows [B][H4][W4][128]
  int B, H, W, H2, W2, H4, W4, tiles_y, tiles_x;
};

constexpr int PTH = 8, PTW = 16;                    // pooled tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1; // conv outputs behind it: 17 x 33
constexpr int CPIX = CTH * CTW;                     // 561
constexpr int CBLK = (CPIX + 15) / 16;              // 36 row blocks
constexpr int PH = CTH + 3, PW = CTW + 3;           // input patch: 20 x 36 space-to-depth pixels of 32 B
constexpr int NPATCH = PH * PW;                     // 720
constexpr int OFF_PH = 0, OFF_PL = 23552;           // heads / tails patch (23 040 B each)
constexpr int OFF_WH = 47104, OFF_WL = OFF_WH + 32768;   // [4 filter rows][64][128 B] each
constexpr int OFF_CONV = 0;                         // fp32 [576][256 B], over the patches and the filter once the conv is done
constexpr int OFF_VEC = CBLK * 16 * 256;            // 147 456: scale, shift
constexpr int LDS_BYTES = OFF_VEC + 128 * 4;        // 147 968
static_assert(OFF_WL + 32768 <= OFF_VEC, "lds map");

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
// filter images: rows of 128 B, the key of stem.hip (fragment lanes read rows {0-3, 8-11, 16-19, 24-27} (+4): one more row bit in the XOR)
__device__ __forceinline__ int wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1)) << 4); }
__device__ __forceinline__ int wrow(int j, int lr) { return (j >> 1) * 32 + (lr >> 2) * 8 + (j & 1) * 4 + (lr & 3); }
// conv tile: rows of 256 B (64 fp32), 16-B chunks XORed with the row
__device__ __forceinline__ int cswz(int row, int c16) { return row * 256 + ((c16 ^ (row & 15)) << 4); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_pool_x3_kernel(const StemX3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = p.tiles_y * p.tiles_x, total = p.B * ntile;
  // PERSISTENT: a workgroup walks a contiguous range of tiles (neighbours share their halo lines in this XCD's L2); the next tile's image
  // pixels are loaded into registers before the current tile's conv, so the image's memory latency is exposed once per workgroup.  (The
  // filter comes back from L2 by LDS-DMA for every tile -- the conv tile overlays it; kept in registers, 8 x 16 B per thread, the kernel spills.)
  const int q0 = (int)((long long)blockIdx.x * total / gridDim.x), q1 = (int)((long long)(blockIdx.x + 1) * total / gridDim.x);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 64 * 16 * 64 * 2, 0x00020000);

  float* const vec = reinterpret_cast<float*>(smem + OFF_VEC);
  if (t < 64) vec[t] = p.scale[t];
  else if (t < 128) vec[t] = p.shift[t - 64];

  // ---- filter by LDS-DMA: LDS row n of filter row R = 4 taps x 16 slots; chunk kc = 16 B = slots 8 (kc & 1) .. + 7 of tap kc >> 1
  const int fdrow = lane >> 3;
  const int fkc = (lane & 7) ^ ((4 * (uw & 1) + (lane >> 4)) & 7) ^ (((uw >> 1) & 1) << 1);      // wswz key of rows 8 uw + drow
  const unsigned foff = (unsigned)((((8 * uw + fdrow) * 16 + (fkc >> 1)) * 64 + (fkc & 1) * 8) * 2);
  // ---- image patch of a tile: one space-to-depth pixel per item = 2 x 2 image pixels x 3 channels, two items per thread
  float2 pf[2][6];
  auto load_img = [&](int wg) {
    const int b = wg / ntile, tt = wg - b * ntile;
    const int iy0 = 2 * ((tt / p.tiles_x) * PTH) - 3, ix0 = 2 * ((tt % p.tiles_x) * PTW) - 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = it * 512 + t;
      const int pr = q / PW, pc = q - pr * PW;
      const int Y = iy0 + pr, X = ix0 + pc;
      const bool ok = q < NPATCH && (unsigned)Y < (unsigned)p.H2 && (unsigned)X < (unsigned)p.W2;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          float2 f = make_float2(0.f, 0.f);
          if (ok) f = *reinterpret_cast<const float2*>(p.img + (((long long)b * 3 + c) * p.H + 2 * Y + dy) * (long long)p.W + 2 * X);
          pf[it][c * 2 + dy] = f;
        }
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = it * 512 + t;
      if (q < NPATCH) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            v[(dy * 2 + 0) * 3 + c] = pf[it][c * 2 + dy].x;
            v[(dy * 2 + 1) * 3 + c] = pf[it][c * 2 + dy].y;
          }
        bf16x8 h0, h1, l0, l1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          h0[k] = (bf16_t)v[k]; l0[k] = (bf16_t)(v[k] - (float)h0[k]);
          h1[k] = (bf16_t)v[8 + k]; l1[k] = (bf16_t)(v[8 + k] - (float)h1[k]);
        }
        *reinterpret_cast<bf16x8*>(smem + OFF_PH + q * 32) = h0; *reinterpret_cast<bf16x8*>(smem + OFF_PH + q * 32 + 16) = h1;
        *reinterpret_cast<bf16x8*>(smem + OFF_PL + q * 32) = l0; *reinterpret_cast<bf16x8*>(smem + OFF_PL + q * 32 + 16) = l1;
      }
    }
  };
  if (q0 < q1) load_img(q0);
  for (int wg = q0; wg < q1; ++wg) {
  const int b = wg / ntile, tt = wg - b * ntile;
  const int py0 = (tt / p.tiles_x) * PTH, px0 = (tt % p.tiles_x) * PTW;   // first pooled pixel of the tile
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;                          // first conv output behind it
  __syncthreads();          // the previous tile's pool pass is done with the conv tile (first tile: scale / shift are in place)
#pragma unroll
  for (int R = 0; R < 4; ++R) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + OFF_WH + R * 8192 + uw * 1024), 16, foff + R * 512u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + OFF_WL + R * 8192 + uw * 1024), 16, foff + R * 512u + 64u, 0, 0, 0);
  }
  write_patch();
  __syncthreads();
  if (wg + 1 < q1) load_img(wg + 1);      // lands under this tile's conv

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
    abase[i] = (oy * PW + ox) * 32 + lq * 16;
  }
#pragma unroll
  for (int R = 0; R < 4; ++R) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wh[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = R * 8192 + wswz(wrow(j, lr), ks * 4 + lq);
        wh[j] = *reinterpret_cast<const bf16x8*>(smem + OFF_WH + o);
        wl[j] = *reinterpret_cast<const bf16x8*>(smem + OFF_WL + o);
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (uw + 8 * i < CBLK) {
          const int o = abase[i] + R * (PW * 32) + ks * 64;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(smem + OFF_PH + o), al = *reinterpret_cast<const bf16x8*>(smem + OFF_PL + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();          // every wave is done with the patches and the filter: the conv tile takes their place

  // ---- 3a. BN + ReLU, head + tail rounding, fp32 conv tile
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
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(vec + c + 4 * q), sh = *reinterpret_cast<const f32x4*>(vec + 64 + c + 4 * q);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = ok ? fmaxf(acc[i][2 * jp + q][r] * sc[r] + sh[r], 0.f) : 0.f;
            *reinterpret_cast<f32x4*>(ct + cswz(pidx, (c >> 2) + q)) = o;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- 3b. 3 x 3 / stride-2 max over the conv tile; item = (pooled pixel, octet of channels)
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
        const f32x4 a = *reinterpret_cast<const f32x4*>(ct + cswz(row, 2 * c8)), c = *reinterpret_cast<const f32x4*>(ct + cswz(row, 2 * c8 + 1));
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = fmaxf(m[k], a[k]); m[4 + k] = fmaxf(m[4 + k], c[k]); }
      }
    const int py = py0 + ppy, px = px0 + ppx;
    if (py < p.H4 && px < p.W4) {
      bf16x8 h, l;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // the conv output as its X-layout store would hold it (head + tail), THEN the pooled value's own head and tail: rounding to
        // head + tail is monotone, so the max of the rounded values is the rounded max -- one rounding per pooled value instead of 4.4
        const bf16_t ch = (bf16_t)m[k];
        const float v = (float)ch + (float)(bf16_t)(m[k] - (float)ch);
        h[k] = (bf16_t)v; l[k] = (bf16_t)(v - (float)h[k]);
      }
      bf16_t* dst = p.y + (((long long)b * p.H4 + py) * p.W4 + px) * 128 + ((c8 >> 2) << 6) + ((c8 & 3) << 3);
      *reinterpret_cast<bf16x8*>(dst) = h;
      *reinterpret_cast<bf16x8*>(dst + 32) = l;
    }
  }
  }      // tiles of this workgroup
}

}  // namespace

extern "C" int aod_stem_pool_x3_fwd(const float* img, const void* w_x, const float* scale, const float* shift, void* y, int B, int C, int H, int W,
                                    aod_stream_t stream) {
  AOD_CHECK_ARG(img && w_x && scale && shift && y && B >= 1, "stem_pool_x3: bad args");
  AOD_CHECK_ARG(C == 3 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "stem_pool_x3: a 3-channel image with even sides (C %d, %d x %d)", C, H, W);
  StemX3Args a;
  a.img = img; a.w = (const bf16_t*)w_x; a.scale = scale; a.shift = shift; a.y = (bf16_t*)y;
  a.B = B; a.H = H; a.W = W; a.H2 = H / 2; a.W2 = W / 2;
  a.H4 = (a.H2 - 1) / 2 + 1; a.W4 = (a.W2 - 1) / 2 + 1;          // max_pool2d(kernel 3, stride 2, padding 1)
  a.tiles_y = (a.H4 + PTH - 1) / PTH; a.tiles_x = (a.W4 + PTW - 1) / PTW;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  const int total = B * a.tiles_y * a.tiles_x;
  hipLaunchKernelGGL(stem_pool_x3_kernel, dim3(total < 256 ? total : 256), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  AOD_LAUNCH_CHECK();
  return 0;
}
