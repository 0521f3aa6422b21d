// Kernels specific to the SSD300-VGG16 + MEH variant (BASELINE config 0; reference: backbones/ssd_vgg.py,
// necks/ssd_neck.py:105-128 L2Norm, dense_heads/My_L_ssd_head.py:182-224 losses).  Small tensors (300x300 inputs):
// simple one-thread-per-output kernels; NHWC bf16 activations, fp32 logits.
#include "common.h"

static inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// ---------------------------------------------------------------- generic max-pool (k x k, stride s, pad p), NHWC bf16
// torch semantics: windows clipped to the image, -inf padding, first maximum in (dy, dx) scan order wins the gradient.
__global__ void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C8, int OH, int OW, int k, int s, int p) {
  const long long n = (long long)B * OH * OW * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % C8; long long r = i / C8;
    const int ox = r % OW; r /= OW;
    const int oy = r % OH; const int b = r / OH;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
      const int yy = oy * s - p + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int xx = ox * s - p + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((((long long)b * H + yy) * W + xx) * C8 + c) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)m[j];
    *reinterpret_cast<bf16x8*>(y + i * 8) = o;
  }
}
// backward as a gather: input pixel (yy, xx) receives g[oy, ox] from every window whose FIRST maximum it is
__global__ void maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g, bf16_t* __restrict__ gx, int B, int H, int W, int C8,
                                   int OH, int OW, int k, int s, int p) {
  const long long n = (long long)B * H * W * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = i % C8; long long r = i / C8;
    const int xx = r % W; r /= W;
    const int yy = r % H; const int b = r / H;
    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + i * 8);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // windows containing (yy, xx): oy in [ceil((yy + p - k + 1)/s), floor((yy + p)/s)]
    int oy0 = yy + p - k + 1; oy0 = oy0 <= 0 ? 0 : (oy0 + s - 1) / s;
    int ox0 = xx + p - k + 1; ox0 = ox0 <= 0 ? 0 : (ox0 + s - 1) / s;
    const int oy1 = min((yy + p) / s, OH - 1), ox1 = min((xx + p) / s, OW - 1);
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        // is (yy, xx) the first maximum of window (oy, ox), per channel?
        bool win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = true;
        for (int dy = 0; dy < k; ++dy) {
          const int y2 = oy * s - p + dy;
          if ((unsigned)y2 >= (unsigned)H) continue;
          for (int dx = 0; dx < k; ++dx) {
            const int x2 = ox * s - p + dx;
            if ((unsigned)x2 >= (unsigned)W || (y2 == yy && x2 == xx)) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((((long long)b * H + y2) * W + x2) * C8 + c) * 8);
            const bool before = (y2 < yy) || (y2 == yy && x2 < xx);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float a = (float)v[j], me = (float)xv[j];
              if (a > me || (before && a == me)) win[j] = false;
            }
          }
        }
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + ((((long long)b * OH + oy) * OW + ox) * C8 + c) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (win[j]) acc[j] += (float)gv[j];
      }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
    *reinterpret_cast<bf16x8*>(gx + i * 8) = o;
  }
}
extern "C" int aod_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k, int s, int p, aod_stream_t stream) {
  AOD_CHECK_ARG(x && y && C % 8 == 0 && k >= 1 && s >= 1, "maxpool_fwd: bad args");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)B * OH * OW * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, B, H, W, C / 8, OH, OW, k, s, p);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_maxpool_bwd(const void* x, const void* g, void* gx, int B, int H, int W, int C, int OH, int OW, int k, int s, int p,
                               aod_stream_t stream) {
  AOD_CHECK_ARG(x && g && gx && C % 8 == 0, "maxpool_bwd: bad args");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)g, (bf16_t*)gx, B, H, W, C / 8, OH, OW, k, s, p);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- L2Norm (ssd_neck.py:105-128): y = w * x / (||x||_2 + eps) per pixel
// one wavefront per pixel, channels strided over lanes
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y, long long rows,
                                                         int C, float eps) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = (float)x[r * C + c]; ss += v * v; }
  ss = wave_sum(ss);
  const float n = sqrtf(ss) + eps;
  for (int c = lane; c < C; c += 64) y[r * C + c] = (bf16_t)(w[c] * (float)x[r * C + c] / n);
}
// (cs_ws: deterministic mode -- the four rows of a block are added in wave order through LDS and stored as ONE partial row per block, which
// aod_colsum_finalize adds in block order; C <= 1024 there)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ g,
                                                         bf16_t* __restrict__ gx, float* __restrict__ gw, long long rows, int C, float eps,
                                                         float* __restrict__ cs_ws) {
  __shared__ float sm[4][1024];
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (cs_ws) {
    const int wv = threadIdx.x >> 6;
    float n = 1.f, k = 0.f;
    if (r < rows) {
      float ss = 0.f, dot = 0.f;
      for (int c = lane; c < C; c += 64) { const float v = (float)x[r * C + c]; ss += v * v; dot += w[c] * (float)g[r * C + c] * v; }
      ss = wave_sum(ss); dot = wave_sum(dot);
      const float nrm = sqrtf(ss);
      n = nrm + eps; k = nrm > 0.f ? dot / (n * n * nrm) : 0.f;
    }
    for (int c = lane; c < C; c += 64) {
      float contrib = 0.f;
      if (r < rows) {
        const float v = (float)x[r * C + c], gg = (float)g[r * C + c];
        gx[r * C + c] = (bf16_t)(w[c] * gg / n - k * v);
        contrib = gg * v / n;
      }
      sm[wv][c] = contrib;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) cs_ws[(long long)blockIdx.x * C + c] = ((sm[0][c] + sm[1][c]) + sm[2][c]) + sm[3][c];
    return;
  }
  if (r >= rows) return;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = (float)x[r * C + c];
    ss += v * v;
    dot += w[c] * (float)g[r * C + c] * v;
  }
  ss = wave_sum(ss); dot = wave_sum(dot);
  const float nrm = sqrtf(ss), n = nrm + eps;
  const float k = nrm > 0.f ? dot / (n * n * nrm) : 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = (float)x[r * C + c], gg = (float)g[r * C + c];
    gx[r * C + c] = (bf16_t)(w[c] * gg / n - k * v);
    atomicAdd(gw + c, gg * v / n);
  }
}
extern "C" int aod_l2norm_fwd(const void* x, const float* w, void* y, int64_t rows, int C, float eps, aod_stream_t stream) {
  if (rows == 0) return 0;
  AOD_CHECK_ARG(x && w && y, "l2norm_fwd: null");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)y, (long long)rows, C, eps);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_l2norm_bwd(const void* x, const float* w, const void* g, void* gx, float* gw, int64_t rows, int C, float eps, aod_stream_t stream) {
  if (rows == 0) return 0;
  AOD_CHECK_ARG(x && w && g && gx && gw, "l2norm_bwd: null");
  const int nb = (int)((rows + 3) / 4);
  float* const cs_ws = C <= 1024 ? aod_det_scratch((size_t)nb * C) : nullptr;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (const bf16_t*)g, (bf16_t*)gx,
                     gw, (long long)rows, C, eps, cs_ws);
  AOD_LAUNCH_CHECK();
  if (cs_ws) return aod_colsum_finalize(cs_ws, nb, C, C, gw, nullptr, 0, (hipStream_t)stream);
  return 0;
}

// ---------------------------------------------------------------- SSD loss (My_L_ssd_head.py:182-215), one block per image
// ce[a] = CE(logits[a], label[a]) * label_w[a];  loss_cls = (sum_pos ce + sum of the min(3*#pos, #neg) largest negative ce);
// loss_bbox = sum smooth_l1(pred - tgt, beta) * w.  Outputs per image: ce[A] (== loss_noR), sums[b] = (cls_sum, bbox_sum, mean(ce)),
// thr[b] = (k-th largest negative ce bits, number of ties at the threshold that are selected, #pos, k).
constexpr int SB = 1024;
__global__ __launch_bounds__(SB) void ssd_loss_fwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels, const float* __restrict__ lw,
                                                          const float* __restrict__ bp, const float* __restrict__ bt, const float* __restrict__ bw,
                                                          int A, int C1, int num_classes, int neg_pos_ratio, float beta, float* __restrict__ ce_out,
                                                          float* __restrict__ sums, unsigned* __restrict__ sel) {
  __shared__ int hist[256];
  __shared__ float red[SB / 64];
  __shared__ int redi[SB / 64];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* x = cls + (long long)b * A * C1;
  const long long* lab = labels + (long long)b * A;
  float* ce = ce_out + (long long)b * A;
  float pos_sum = 0.f, box_sum = 0.f, all_sum = 0.f;
  int npos = 0, nneg = 0;
  for (int a = t; a < A; a += SB) {
    const float* r = x + (long long)a * C1;
    float m = r[0];
    for (int c = 1; c < C1; ++c) m = fmaxf(m, r[c]);
    float s = 0.f;
    for (int c = 0; c < C1; ++c) s += expf(r[c] - m);
    const long long l = lab[a];
    const float v = (logf(s) + m - r[l]) * lw[(long long)b * A + a];     // -log_softmax[label] * weight
    ce[a] = v;
    all_sum += v;
    if (l >= 0 && l < num_classes) { pos_sum += v; ++npos; }
    else if (l == num_classes) ++nneg;
    for (int j = 0; j < 4; ++j) {
      const long long o = ((long long)b * A + a) * 4 + j;
      const float d = fabsf(bp[o] - bt[o]);
      box_sum += (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) * bw[o];
    }
  }
  // block reductions
  auto bsum = [&](float v) { v = wave_sum(v); __syncthreads(); if ((t & 63) == 0) red[t >> 6] = v; __syncthreads(); float r = 0.f; for (int i = 0; i < SB / 64; ++i) r += red[i]; return r; };
  auto bsumi = [&](int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); __syncthreads(); if ((t & 63) == 0) redi[t >> 6] = v; __syncthreads(); int r = 0; for (int i = 0; i < SB / 64; ++i) r += redi[i]; return r; };
  pos_sum = bsum(pos_sum); box_sum = bsum(box_sum); all_sum = bsum(all_sum);
  npos = bsumi(npos); nneg = bsumi(nneg);
  int k = neg_pos_ratio * npos;
  if (k > nneg) k = nneg;
  // k-th largest negative ce by 4-pass radix select on the float bits (ce >= 0)
  unsigned prefix = 0u, mask = 0u;
  int remaining = k;
  if (k > 0) {
    for (int pass = 3; pass >= 0; --pass) {
      for (int i = t; i < 256; i += SB) hist[i] = 0;
      __syncthreads();
      const int shift = pass * 8;
      for (int a = t; a < A; a += SB)
        if (lab[a] == num_classes) {
          const unsigned u = __float_as_uint(fmaxf(ce[a], 0.f));
          if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 0xffu], 1);
        }
      __syncthreads();
      int d = 255, acc = 0;
      for (; d >= 0; --d) { const int h = hist[d]; if (acc + h >= remaining) break; acc += h; }
      remaining -= acc;
      prefix |= ((unsigned)d) << shift;
      mask |= 0xffu << shift;
      __syncthreads();
    }
  }
  // sum of negatives strictly above the threshold + `remaining` copies of the threshold value
  float neg_sum = 0.f;
  if (k > 0)
    for (int a = t; a < A; a += SB)
      if (lab[a] == num_classes && __float_as_uint(fmaxf(ce[a], 0.f)) > prefix) neg_sum += ce[a];
  neg_sum = bsum(neg_sum);
  if (t == 0) {
    if (k > 0) neg_sum += (float)remaining * __uint_as_float(prefix);
    sums[b * 3 + 0] = pos_sum + neg_sum;
    sums[b * 3 + 1] = box_sum;
    sums[b * 3 + 2] = all_sum / (float)A;
    sel[b * 4 + 0] = prefix; sel[b * 4 + 1] = (unsigned)remaining; sel[b * 4 + 2] = (unsigned)npos; sel[b * 4 + 3] = (unsigned)k;
  }
}
// backward: d/dlogits of [ g_cls[b] * (selected ce) + g_noR[b][a] * ce[a] ], d/dpred of g_box[b] * smooth_l1
__global__ __launch_bounds__(SB) void ssd_loss_bwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels, const float* __restrict__ lw,
                                                          const float* __restrict__ bp, const float* __restrict__ bt, const float* __restrict__ bw,
                                                          const float* __restrict__ ce, const unsigned* __restrict__ sel, int A, int C1,
                                                          int num_classes, float beta, const float* __restrict__ g_cls, const float* __restrict__ g_box,
                                                          const float* __restrict__ g_noR, float* __restrict__ gcls, float* __restrict__ gbox) {
  __shared__ int s_tie;
  const int b = blockIdx.x, t = threadIdx.x;
  const unsigned thr = sel[b * 4 + 0];
  const int nties = (int)sel[b * 4 + 1], k = (int)sel[b * 4 + 3];
  if (t == 0) s_tie = 0;
  __syncthreads();
  const float gc = g_cls ? g_cls[b] : 0.f, gb = g_box ? g_box[b] : 0.f;
  // ties at the threshold: the first `nties` in index order are selected (one pass, ordered by a serial scan of thread 0 is avoided by
  // letting each tie claim a ticket in index order per 1024-chunk)
  for (int a0 = 0; a0 < A; a0 += SB) {
    const int a = a0 + t;
    bool tie = false, selneg = false, ispos = false;
    float coef = 0.f;
    long long l = 0;
    if (a < A) {
      l = labels[(long long)b * A + a];
      const unsigned u = __float_as_uint(fmaxf(ce[(long long)b * A + a], 0.f));
      ispos = l >= 0 && l < num_classes;
      if (l == num_classes && k > 0) { selneg = u > thr; tie = (u == thr); }
    }
    // ordered ticketing of ties inside this chunk
    const unsigned long long m = __ballot(tie);
    const int lane = t & 63, w = t >> 6;
    __shared__ int wcnt[SB / 64];
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int before = s_tie;
    for (int i = 0; i < w; ++i) before += wcnt[i];
    before += __popcll(m & ((1ull << lane) - 1ull));
    if (tie && before < nties) selneg = true;
    __syncthreads();
    if (t == 0) { int tot = 0; for (int i = 0; i < SB / 64; ++i) tot += wcnt[i]; s_tie += tot; }
    __syncthreads();
    if (a < A) {
      const long long row = (long long)b * A + a;
      coef = ((ispos || selneg) ? gc : 0.f) + (g_noR ? g_noR[row] : 0.f);
      coef *= lw[row];
      const float* r = cls + row * C1;
      float mx = r[0];
      for (int c = 1; c < C1; ++c) mx = fmaxf(mx, r[c]);
      float s = 0.f;
      for (int c = 0; c < C1; ++c) s += expf(r[c] - mx);
      for (int c = 0; c < C1; ++c) gcls[row * C1 + c] = coef * (expf(r[c] - mx) / s - (c == l ? 1.f : 0.f));
      for (int j = 0; j < 4; ++j) {
        const float d = bp[row * 4 + j] - bt[row * 4 + j];
        const float ad = fabsf(d);
        const float gsl = ad < beta ? d / beta : (d > 0.f ? 1.f : -1.f);
        gbox[row * 4 + j] = gb * gsl * bw[row * 4 + j];
      }
    }
  }
}
extern "C" int aod_ssd_loss_fwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred, const float* bbox_tgt,
                                const float* bbox_w, int B, int A, int C1, int num_classes, int neg_pos_ratio, float beta, float* ce, float* sums3,
                                uint32_t* sel4, aod_stream_t stream) {
  if (B == 0) return 0;
  AOD_CHECK_ARG(cls && labels && label_w && bbox_pred && bbox_tgt && bbox_w && ce && sums3 && sel4 && C1 >= 2, "ssd_loss_fwd: bad args");
  hipLaunchKernelGGL(ssd_loss_fwd_kernel, dim3(B), dim3(SB), 0, (hipStream_t)stream, cls, (const long long*)labels, label_w, bbox_pred, bbox_tgt, bbox_w, A, C1,
                     num_classes, neg_pos_ratio, beta, ce, sums3, sel4);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_ssd_loss_bwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred, const float* bbox_tgt,
                                const float* bbox_w, const float* ce, const uint32_t* sel4, int B, int A, int C1, int num_classes, float beta,
                                const float* g_cls, const float* g_box, const float* g_noR, float* grad_cls, float* grad_box, aod_stream_t stream) {
  if (B == 0) return 0;
  AOD_CHECK_ARG(cls && labels && label_w && bbox_pred && bbox_tgt && bbox_w && ce && sel4 && grad_cls && grad_box, "ssd_loss_bwd: bad args");
  hipLaunchKernelGGL(ssd_loss_bwd_kernel, dim3(B), dim3(SB), 0, (hipStream_t)stream, cls, (const long long*)labels, label_w, bbox_pred, bbox_tgt, bbox_w, ce, sel4, A,
                     C1, num_classes, beta, g_cls, g_box, g_noR, grad_cls, grad_box);
  AOD_LAUNCH_CHECK();
  return 0;
}
