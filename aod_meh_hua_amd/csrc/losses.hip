// Fused EDL softmax-focal + L1 loss (forward / backward) and MEH loss for gfx950.
// HBM-bound: one pass over the fp32 logits.  Rows are staged through LDS with coalesced 16-B
// loads, then one thread owns one anchor row (C = 20 logits) in registers.
// Compiled with -ffp-contract=off: the formulas follow the reference op order.
#include "common.h"

#define FLT_MIN_F 1.17549435e-38f
constexpr int LB = 256;     // rows per block
constexpr int MAXC = 96;

__device__ __forceinline__ float focal_pow(float b, float gamma) { return gamma == 2.f ? b * b : powf(b, gamma); }

// The row kernels are VALU-bound, not HBM-bound: per class the reference formula costs two exponentials, up to three logarithms and three
// divisions, and libm's expf / logf / IEEE division are 12-30 instructions each (~4000 instructions per 20-class row; PMC and the
// secondary roofline of bench.py: 0.10 of the HBM rate).  AOD_LOSS_FAST_MATH (default) evaluates them with the hardware transcendentals
// (v_exp_f32, v_log_f32, v_rcp_f32: 1 ulp each, arguments are normal floats), same formulas in the same order.  The golden parity tests
// keep their tolerances (rtol 2e-5 per row against the reference's own values); -DAOD_LOSS_FAST_MATH=0 restores libm.
#ifndef AOD_LOSS_FAST_MATH
#define AOD_LOSS_FAST_MATH 1
#endif
#if AOD_LOSS_FAST_MATH
__device__ __forceinline__ float l_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float l_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float l_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
__device__ __forceinline__ float l_exp(float x) { return expf(x); }
__device__ __forceinline__ float l_log(float x) { return logf(x); }
__device__ __forceinline__ float l_div(float a, float b) { return a / b; }
#endif

// Row layout in the kernels below: LPR lanes share one anchor row, lane part j owns the classes [j*cpl, (j+1)*cpl) with
// cpl = ceil(C / LPR) <= CT.  CT is a compile-time bound: the loops are fully unrolled and predicated, so x[] / p[] / gp[] live in
// registers -- with a run-time trip count (or a 96-wide bound for the 80/81-class heads) they were demoted to scratch memory
// (400-1168 B per lane) and the kernels ran at an eighth of the HBM rate.  LPR = 1 (C <= 24: VOC's 20/21 classes, one lane per row, the
// reference's sequential op order) or 4 (C <= 96: COCO's 80/81; row maxima / sums are combined over the 4 lanes by two shuffles).
template <int LPR>
__device__ __forceinline__ float grp_max(float v) {
  if (LPR >= 2) v = fmaxf(v, __shfl_xor(v, 1, 64));
  if (LPR >= 4) v = fmaxf(v, __shfl_xor(v, 2, 64));
  return v;
}
template <int LPR>
__device__ __forceinline__ float grp_sum(float v) {
  if (LPR >= 2) v += __shfl_xor(v, 1, 64);
  if (LPR >= 4) v += __shfl_xor(v, 2, 64);
  return v;
}

// per-row forward core: fills p[] (softmax) and returns the row loss (identical in the LPR lanes of the row)
template <int CT, int LPR>
__device__ __forceinline__ float edl_row_fwd(const float* x, int n, int c0, long long label, float gamma, float alpha, float* p) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < n) m = fmaxf(m, x[c]);
  m = grp_max<LPR>(m);
  float S = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < n) { p[c] = l_exp(x[c] - m); S += p[c]; }
  S = grp_sum<LPR>(S);
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < n) {
    const float pr = l_div(p[c], S);
    p[c] = pr;
    const float z = l_log(l_div(pr, 1.f - pr + 1e-9f) + 1e-9f);
    const float q = l_div(1.f, 1.f + l_exp(-z));
    const bool pos = label == c0 + c;
    // (one logarithm per class: its argument is selected, not its result)
    const float lg = l_log(fmaxf(pos ? q : 1.f - q, FLT_MIN_F));
    float l;
    if (pos) l = -alpha * focal_pow(1.f - q, gamma) * lg;
    else l = -(1.f - alpha) * focal_pow(q, gamma) * lg;
    tot += l;
  }
  return grp_sum<LPR>(tot);
}

template <int CT, int LPR>
__global__ __launch_bounds__(LB) void edl_l1_fwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels,
                                                        const float* __restrict__ lw, const float* __restrict__ bp,
                                                        const float* __restrict__ bt, const float* __restrict__ bw, long long nrows, int C,
                                                        float gamma, float alpha, float* __restrict__ loss_noR, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  constexpr int RB = LB / LPR;   // rows per block
  const int P = C | 1;  // odd pitch -> conflict-free row reads
  const long long r0 = (long long)blockIdx.x * RB;
  const int nr = (int)min((long long)RB, nrows - r0);
  aod_stage_rows<LB>(cls + r0 * C, srow, nr, C, P);
  __syncthreads();
  float s_cls = 0.f, s_box = 0.f, s_nor = 0.f;
  const int lrow = threadIdx.x / LPR, part = threadIdx.x % LPR;
  const bool live = lrow < nr;
  const int row = live ? lrow : nr - 1;          // every lane takes part in the row shuffles
  const int cpl = (C + LPR - 1) / LPR, c0 = part * cpl;
  const int n = min(cpl, C - c0);
  {
    const long long r = r0 + row;
    float x[CT], p[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) x[c] = c < n ? srow[row * P + c0 + c] : 0.f;
    const float l = edl_row_fwd<CT, LPR>(x, n, c0, labels[r], gamma, alpha, p);
    if (live && part == 0) {
      loss_noR[r] = l;
      s_nor = l;
      s_cls = l * lw[r];
      if (bp) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(bp + r * 4), b = *reinterpret_cast<const f32x4*>(bt + r * 4),
                    w = *reinterpret_cast<const f32x4*>(bw + r * 4);
        for (int j = 0; j < 4; ++j) s_box += fabsf(a[j] - b[j]) * w[j];
      }
    }
  }
  // block reduction (fixed order -> deterministic)
  __shared__ float red[3][LB / 64];
  s_cls = wave_sum(s_cls); s_box = wave_sum(s_box); s_nor = wave_sum(s_nor);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_cls; red[1][threadIdx.x >> 6] = s_box; red[2][threadIdx.x >> 6] = s_nor; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = 0.f;
    for (int i = 0; i < LB / 64; ++i) v += red[threadIdx.x][i];
    partials[(long long)blockIdx.x * 3 + threadIdx.x] = v;
  }
}

// second stage: one block sums the per-block partials in a fixed order and ADDS into out[k]
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, long long nblocks, int k, float* __restrict__ out) {
  __shared__ float red[4];
  for (int j = 0; j < k; ++j) {
    float v = 0.f;
    for (long long i = threadIdx.x; i < nblocks; i += 256) v += partials[i * k + j];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[j] += red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}

static inline int edl_lpr(int C) { return C <= 24 ? 1 : 4; }
static inline long long edl_blocks(int64_t nrows, int C) { const int rb = LB / edl_lpr(C); return (nrows + rb - 1) / rb; }

// (sized for the LPR = 4 form: 64 rows per block)
extern "C" size_t aod_loss_partials_len(int64_t nrows) { return (size_t)((nrows + LB / 4 - 1) / (LB / 4)) * 3; }

extern "C" int aod_edl_focal_l1_fwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred,
                                    const float* bbox_tgt, const float* bbox_w, int64_t nrows, int C, float gamma, float alpha,
                                    float* loss_noR, float* sums3, float* partials, aod_stream_t stream) {
  if (nrows == 0) return 0;
  AOD_CHECK_ARG(cls && labels && label_w && loss_noR && sums3 && partials, "edl_fwd: null pointer");
  AOD_CHECK_ARG(C >= 1 && C <= MAXC, "edl_fwd: C=%d out of range", C);
  AOD_CHECK_ARG(!bbox_pred || (bbox_tgt && bbox_w), "edl_fwd: bbox_pred needs targets and weights");
  const long long nb = edl_blocks(nrows, C);
  if (C <= 24)
    hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 1>), dim3((unsigned)nb), dim3(LB), (size_t)LB * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                       label_w, bbox_pred, bbox_tgt, bbox_w, (long long)nrows, C, gamma, alpha, loss_noR, partials);
  else
    hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 4>), dim3((unsigned)nb), dim3(LB), (size_t)(LB / 4) * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                       label_w, bbox_pred, bbox_tgt, bbox_w, (long long)nrows, C, gamma, alpha, loss_noR, partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, 3, sums3);
  AOD_LAUNCH_CHECK();
  return 0;
}

// backward.  Output element (row r, class c) lives at (r / A) * pitch + (r % A) * C + c so that the
// gradient lands directly in the conv's [pixels, A*C (padded)] dZ layout, bf16 or fp32.
template <bool OUT_BF16, int CT, int LPR>
__global__ __launch_bounds__(LB) void edl_l1_bwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels,
                                                        const float* __restrict__ lw, const float* __restrict__ bp,
                                                        const float* __restrict__ bt, const float* __restrict__ bw, long long nrows, int C,
                                                        float gamma, float alpha, const float* __restrict__ g_cls,
                                                        const float* __restrict__ g_bbox, const float* __restrict__ g_noR, float g_noR_s, int g_noR_bcast,
                                                        void* __restrict__ grad_cls, void* __restrict__ grad_bbox, int A, int pitch_cls,
                                                        int pitch_box) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  constexpr int RB = LB / LPR;
  const int P = C | 1;
  const long long r0 = (long long)blockIdx.x * RB;
  const int nr = (int)min((long long)RB, nrows - r0);
  const int tot = nr * C;
  aod_stage_rows<LB>(cls + r0 * C, srow, nr, C, P);
  __shared__ long long s_obase[LB];       // element offset of each row's class 0 in the destination
  __syncthreads();
  const int lrow = threadIdx.x / LPR, part = threadIdx.x % LPR;
  const bool live = lrow < nr;
  const int row = live ? lrow : nr - 1;
  const int cpl = (C + LPR - 1) / LPR, c0 = part * cpl;
  const int n = min(cpl, C - c0);
  {
    const long long r = r0 + row;
    if (live && part == 0) s_obase[row] = (r / A) * pitch_cls + (r % A) * C;
    float x[CT], p[CT], gp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) x[c] = c < n ? srow[row * P + c0 + c] : 0.f;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < n) m = fmaxf(m, x[c]);
    m = grp_max<LPR>(m);
    float S = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < n) { p[c] = l_exp(x[c] - m); S += p[c]; }
    S = grp_sum<LPR>(S);
    const long long label = labels[r];
    const float coef = g_cls[0] * lw[r] + (g_noR ? g_noR[g_noR_bcast ? 0 : r] : g_noR_s);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < n) {
      const float pr = l_div(p[c], S);
      p[c] = pr;
      const float om = 1.f - pr + 1e-9f;
      const float u = l_div(pr, om);
      const float z = l_log(u + 1e-9f);
      const float q = l_div(1.f, 1.f + l_exp(-z));
      const bool pos = label == c0 + c;
      const float lg = l_log(fmaxf(pos ? q : 1.f - q, FLT_MIN_F));
      float gz;   // d l / d z  (mmcv sigmoid_focal_loss backward)
      if (pos) gz = -alpha * focal_pow(1.f - q, gamma) * (1.f - q - gamma * q * lg);
      else gz = -(1.f - alpha) * focal_pow(q, gamma) * (gamma * (1.f - q) * lg - q);
      // dz/dp = (1+eps) / ((1-p+eps)^2 (u+eps))
      const float g = l_div(coef * gz * (1.f + 1e-9f), om * om * (u + 1e-9f));
      gp[c] = g;
      dot += pr * g;
    }
    dot = grp_sum<LPR>(dot);
    // the row's gradient goes back into its LDS row; the block then stores all rows with consecutive lanes on consecutive elements
    // (one thread storing its own 20 values writes 4 B per lane at an 80-B lane stride)
    if (live) {
#pragma unroll
      for (int c = 0; c < CT; ++c) if (c < n) srow[row * P + c0 + c] = p[c] * (gp[c] - dot);
    }
    if (live && part == 0 && bp && grad_bbox) {
      const float gb = g_bbox[0];
      const long long bb = (r / A) * pitch_box + (r % A) * 4;
      for (int j = 0; j < 4; ++j) {
        const float d = bp[r * 4 + j] - bt[r * 4 + j];
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float gx = sgn * bw[r * 4 + j] * gb;
        if (OUT_BF16) ((bf16_t*)grad_bbox)[bb + j] = (bf16_t)gx;
        else ((float*)grad_bbox)[bb + j] = gx;
      }
    }
  }
  __syncthreads();
  {
    const int drow = LB / C, dcol = LB - drow * C;
    int row = (int)threadIdx.x / C, c = (int)threadIdx.x - row * C;
    for (int i = threadIdx.x; i < tot; i += LB) {
      const long long o = s_obase[row] + c;
      const float gx = srow[row * P + c];
      if (OUT_BF16) ((bf16_t*)grad_cls)[o] = (bf16_t)gx;
      else ((float*)grad_cls)[o] = gx;
      row += drow; c += dcol;
      if (c >= C) { c -= C; ++row; }
    }
  }
}

extern "C" int aod_edl_focal_l1_bwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred,
                                    const float* bbox_tgt, const float* bbox_w, int64_t nrows, int C, float gamma, float alpha,
                                    const float* g_cls, const float* g_bbox, const float* g_noR, float g_noR_scalar, int g_noR_is_scalar,
                                    void* grad_cls, void* grad_bbox, int out_bf16, int A, int pitch_cls, int pitch_box, aod_stream_t stream) {
  if (nrows == 0) return 0;
  AOD_CHECK_ARG(cls && labels && label_w && g_cls && grad_cls, "edl_bwd: null pointer");
  AOD_CHECK_ARG(C >= 1 && C <= MAXC && A >= 1 && pitch_cls >= A * C, "edl_bwd: bad C/A/pitch");
  AOD_CHECK_ARG(!grad_bbox || (bbox_pred && bbox_tgt && bbox_w && g_bbox && pitch_box >= A * 4), "edl_bwd: bbox args");
  const long long nb = edl_blocks(nrows, C);
#define AOD_EDL_BWD(BF, LPR_)                                                                                                              \
  hipLaunchKernelGGL((edl_l1_bwd_kernel<BF, 24, LPR_>), dim3((unsigned)nb), dim3(LB), (size_t)(LB / LPR_) * (C | 1) * 4, (hipStream_t)stream, cls, \
                     (const long long*)labels, label_w, bbox_pred, bbox_tgt, bbox_w, (long long)nrows, C, gamma, alpha, g_cls, g_bbox, g_noR, \
                     g_noR_scalar, g_noR_is_scalar, grad_cls, grad_bbox, A, pitch_cls, pitch_box)
  if (out_bf16) { if (C <= 24) AOD_EDL_BWD(true, 1); else AOD_EDL_BWD(true, 4); }
  else { if (C <= 24) AOD_EDL_BWD(false, 1); else AOD_EDL_BWD(false, 4); }
#undef AOD_EDL_BWD
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- MEH loss
__global__ __launch_bounds__(256) void meh_fwd_kernel(const float* __restrict__ lam, const float* __restrict__ loss, const float* __restrict__ bw4,
                                                      long long n, float* __restrict__ partials) {
  float s = 0.f;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float d = fabsf(lam[i] + 1e-9f - loss[i]) * bw4[i * 4];
    s = d * d;
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
template <bool OUT_BF16>
__global__ void meh_bwd_kernel(const float* __restrict__ lam, const float* __restrict__ loss, const float* __restrict__ bw4, long long n,
                               const float* __restrict__ g, void* __restrict__ grad, int A, int pitch) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float w = bw4[i * 4];
    const float v = g[0] * 2.f * w * w * (lam[i] + 1e-9f - loss[i]);
    const long long o = (i / A) * pitch + (i % A);
    if (OUT_BF16) ((bf16_t*)grad)[o] = (bf16_t)v; else ((float*)grad)[o] = v;
  }
}
extern "C" int aod_meh_loss_fwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n, float* out_sum, float* partials,
                                aod_stream_t stream) {
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && out_sum && partials, "meh_fwd: null pointer");
  if (n == 0) return 0;
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(meh_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, (long long)n, partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, 1, out_sum);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_meh_loss_bwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n, const float* g, void* grad_lam,
                                int out_bf16, int A, int pitch, aod_stream_t stream) {
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && g && grad_lam && A >= 1 && pitch >= A, "meh_bwd: bad args");
  if (n == 0) return 0;
  const long long nb = (n + 255) / 256;
  if (out_bf16) hipLaunchKernelGGL((meh_bwd_kernel<true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, (long long)n, g, grad_lam, A, pitch);
  else hipLaunchKernelGGL((meh_bwd_kernel<false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, (long long)n, g, grad_lam, A, pitch);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Elementwise form [rows, C] of the EDL softmax-focal loss: what EDL_Softmax_FocalLoss.forward(reduction='none') returns in the
// reference (EDL_Softmax_FocalLoss.py:51-69 -> mmcv sigmoid_focal_loss(..., 'none')).  Lambda_L2Net.loss_single never needs it (its
// 'none' caller sums over the classes at once, Lambda_L2.py:116: the fused row kernel above) -- this serves stand-alone drop-in callers.
// One lane per row, classes streamed from global memory in three passes (same formulas and order as edl_row_fwd); not a hot kernel.
__device__ __forceinline__ float edl_elem_terms(float pr, bool pos, float gamma, float alpha, float* gz_out) {
  const float om = 1.f - pr + 1e-9f;
  const float u = l_div(pr, om);
  const float z = l_log(u + 1e-9f);
  const float q = l_div(1.f, 1.f + l_exp(-z));
  const float lg = l_log(fmaxf(pos ? q : 1.f - q, FLT_MIN_F));
  if (gz_out) {
    float gz;
    if (pos) gz = -alpha * focal_pow(1.f - q, gamma) * (1.f - q - gamma * q * lg);
    else gz = -(1.f - alpha) * focal_pow(q, gamma) * (gamma * (1.f - q) * lg - q);
    *gz_out = l_div(gz * (1.f + 1e-9f), om * om * (u + 1e-9f));       // d l / d p
  }
  return pos ? -alpha * focal_pow(1.f - q, gamma) * lg : -(1.f - alpha) * focal_pow(q, gamma) * lg;
}

__global__ __launch_bounds__(256) void edl_elem_kernel(const float* __restrict__ cls, const long long* __restrict__ labels, long long nrows, int C,
                                                      float gamma, float alpha, const float* __restrict__ g, float* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  const float* x = cls + r * C;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
  float S = 0.f;
  for (int c = 0; c < C; ++c) S += l_exp(x[c] - m);
  const long long label = labels[r];
  if (!g) {                    // forward: out[r][c] = l_c
    for (int c = 0; c < C; ++c) out[r * C + c] = edl_elem_terms(l_div(l_exp(x[c] - m), S), label == c, gamma, alpha, nullptr);
    return;
  }
  // backward: out[r][k] = p_k * (G_k - sum_c p_c G_c),  G_c = g[r][c] * d l_c / d p_c
  float dot = 0.f;
  for (int c = 0; c < C; ++c) {
    const float pr = l_div(l_exp(x[c] - m), S);
    float dl;
    edl_elem_terms(pr, label == c, gamma, alpha, &dl);
    dot += pr * (g[r * C + c] * dl);
  }
  for (int c = 0; c < C; ++c) {
    const float pr = l_div(l_exp(x[c] - m), S);
    float dl;
    edl_elem_terms(pr, label == c, gamma, alpha, &dl);
    out[r * C + c] = pr * (g[r * C + c] * dl - dot);
  }
}

extern "C" int aod_edl_focal_elem(const float* cls, const int64_t* labels, int64_t nrows, int C, float gamma, float alpha,
                                  const float* grad_out, float* out, aod_stream_t stream) {
  if (nrows == 0) return 0;
  AOD_CHECK_ARG(cls && labels && out, "edl_elem: null pointer");
  AOD_CHECK_ARG(C >= 1, "edl_elem: C=%d out of range", C);
  hipLaunchKernelGGL(edl_elem_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cls, (const long long*)labels,
                     (long long)nrows, C, gamma, alpha, grad_out, out);
  AOD_LAUNCH_CHECK();
  return 0;
}
