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

// One launch over ALL pyramid levels (Lambda_L2.py:105-121 runs loss_single once per level through multi_apply): the levels' anchor rows are
// adjacent row ranges of the same buffers (level-batched prediction convs, level-major targets), and a block belongs to exactly one level --
// level l owns the blocks [blk_end[l-1], blk_end[l]) and the rows [row_end[l-1], row_end[l]), its blocks start at its first row.  Blocks, rows
// per block and the order of every sum are those of a per-level launch: identical bits.  (Unused entries: blk_end = INT_MAX.)
constexpr int MAXLV = 8;
struct LossLevels { int n; int blk_end[MAXLV]; long long row_end[MAXLV]; };
// (constant indices only: a run-time index into the by-value argument would move it to scratch memory)
__device__ __forceinline__ void level_of_block(const LossLevels& lv, int bid, int& level, int& blk0, long long& row0, long long& rend) {
  level = 0; blk0 = 0; row0 = 0; rend = lv.row_end[0];
#pragma unroll
  for (int q = 0; q + 1 < MAXLV; ++q)
    if (bid >= lv.blk_end[q]) { level = q + 1; blk0 = lv.blk_end[q]; row0 = lv.row_end[q]; rend = lv.row_end[q + 1]; }
}
static int fill_levels(LossLevels& lv, int nlevels, const int64_t* level_rows, int rows_per_block, long long& total_rows) {
  lv.n = nlevels;
  long long r = 0, b = 0;
  for (int l = 0; l < MAXLV; ++l) {
    if (l < nlevels) { r += level_rows[l]; b += (level_rows[l] + rows_per_block - 1) / rows_per_block; }
    lv.row_end[l] = r;
    lv.blk_end[l] = l < nlevels ? (int)b : 0x7fffffff;
  }
  total_rows = r;
  return (int)b;
}

template <int CT, int LPR>
__global__ __launch_bounds__(LB) void edl_l1_fwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels,
                                                        const float* __restrict__ lw, const float* __restrict__ bp,
                                                        const float* __restrict__ bt, const float* __restrict__ bw, const LossLevels lv, int C,
                                                        float gamma, float alpha, float* __restrict__ loss_noR, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  constexpr int RB = LB / LPR;   // rows per block
  const int P = C | 1;  // odd pitch -> conflict-free row reads
  int level, blk0; long long row0, rend;
  level_of_block(lv, (int)blockIdx.x, level, blk0, row0, rend);
  const long long r0 = row0 + (long long)((int)blockIdx.x - blk0) * RB;
  const int nr = (int)min((long long)RB, rend - r0);
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

// ... the same for every level of a level-fused launch: block l sums level l's partials (same order) and WRITES out[j * kstride + l * lstride]
// With `num_pos` (per-image positive counts of aod_max_iou_assign): the sums leave as the per-level loss terms of L_anchor_head.py:266-288 /
// SSL_Lambda.py:136-141 -- rows 0, 1 divided by num_total_samples = sum_b max(num_pos[b], 1) (L_anchor_head.py:300-303), row 2 (k = 3) by the
// level's row count (the mean of loss_noR) --, IEEE divisions like the tensor ops they replace; the divisors are kept for the backward pass.
__global__ __launch_bounds__(256) void reduce_partials_levels_kernel(const float* __restrict__ partials, const LossLevels lv, int k, float* __restrict__ out,
                                                                    int lstride, int kstride, const int* __restrict__ num_pos, int nimg,
                                                                    float* __restrict__ divisors, float* __restrict__ num_total) {
  __shared__ float red[4];
  int b0 = 0, b1 = lv.blk_end[0];
  long long r0 = 0, r1 = lv.row_end[0];
#pragma unroll
  for (int q = 0; q + 1 < MAXLV; ++q)
    if ((int)blockIdx.x > q) { b0 = lv.blk_end[q]; b1 = lv.blk_end[q + 1]; r0 = lv.row_end[q]; r1 = lv.row_end[q + 1]; }
  float nt = 1.f;
  if (num_pos && threadIdx.x == 0) {
    long long tot = 0;
    for (int b = 0; b < nimg; ++b) tot += max(num_pos[b], 1);
    nt = (float)tot;
    if (blockIdx.x == 0 && num_total) num_total[0] = nt;
  }
  const long long nblocks = b1 - b0;
  const float* pp = partials + (long long)b0 * k;
  for (int j = 0; j < k; ++j) {
    float v = 0.f;
    for (long long i = threadIdx.x; i < nblocks; i += 256) v += pp[i * k + j];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long o = (long long)j * kstride + (long long)blockIdx.x * lstride;
      const float v = red[0] + red[1] + red[2] + red[3];
      if (num_pos) {
        const float d = j < 2 ? nt : (float)(r1 - r0);
        divisors[o] = d;
        out[o] = v / d;
      } else out[o] = v;
    }
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
  LossLevels lv; long long tot;
  const long long nb = fill_levels(lv, 1, &nrows, LB / edl_lpr(C), tot);
  if (C <= 24)
    hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 1>), dim3((unsigned)nb), dim3(LB), (size_t)LB * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                       label_w, bbox_pred, bbox_tgt, bbox_w, lv, C, gamma, alpha, loss_noR, partials);
  else
    hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 4>), dim3((unsigned)nb), dim3(LB), (size_t)(LB / 4) * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                       label_w, bbox_pred, bbox_tgt, bbox_w, lv, C, gamma, alpha, loss_noR, partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, 3, sums3);
  AOD_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t aod_loss_levels_partials_len(int nlevels, const int64_t* level_rows) {
  size_t n = 0;
  for (int l = 0; l < nlevels; ++l) n += (size_t)((level_rows[l] + LB / 4 - 1) / (LB / 4)) * 3;
  return n;
}

extern "C" int aod_edl_focal_l1_levels_fwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred,
                                           const float* bbox_tgt, const float* bbox_w, int nlevels, const int64_t* level_rows, int C, float gamma,
                                           float alpha, float* loss_noR, float* sums, float* partials, const int32_t* num_pos, int num_images,
                                           float* divisors, float* num_total, aod_stream_t stream) {
  AOD_CHECK_ARG(nlevels >= 1 && nlevels <= MAXLV && level_rows, "edl_levels_fwd: 1..8 levels");
  AOD_CHECK_ARG(!num_pos || (num_images >= 1 && divisors), "edl_levels_fwd: num_pos needs the image count and a divisor buffer");
  AOD_CHECK_ARG(cls && labels && label_w && loss_noR && sums && partials, "edl_levels_fwd: null pointer");
  AOD_CHECK_ARG(C >= 1 && C <= MAXC, "edl_levels_fwd: C=%d out of range", C);
  AOD_CHECK_ARG(!bbox_pred || (bbox_tgt && bbox_w), "edl_levels_fwd: bbox_pred needs targets and weights");
  LossLevels lv; long long tot;
  const int nb = fill_levels(lv, nlevels, level_rows, LB / edl_lpr(C), tot);
  if (nb) {
    if (C <= 24)
      hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 1>), dim3((unsigned)nb), dim3(LB), (size_t)LB * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                         label_w, bbox_pred, bbox_tgt, bbox_w, lv, C, gamma, alpha, loss_noR, partials);
    else
      hipLaunchKernelGGL((edl_l1_fwd_kernel<24, 4>), dim3((unsigned)nb), dim3(LB), (size_t)(LB / 4) * (C | 1) * 4, (hipStream_t)stream, cls, (const long long*)labels,
                         label_w, bbox_pred, bbox_tgt, bbox_w, lv, C, gamma, alpha, loss_noR, partials);
  }
  // sums[3][nlevels]: row 0 = sum l * w, row 1 = sum |d| * bw, row 2 = sum l, one column per level (an empty level: zeros)
  hipLaunchKernelGGL(reduce_partials_levels_kernel, dim3(nlevels), dim3(256), 0, (hipStream_t)stream, partials, lv, 3, sums, 1, nlevels, (const int*)num_pos,
                     num_images, divisors, num_total);
  AOD_LAUNCH_CHECK();
  return 0;
}

// backward.  Output element (row r, class c) lives at (r / A) * pitch + (r % A) * C + c so that the
// gradient lands directly in the conv's [pixels, A*C (padded)] dZ layout, bf16 or fp32.
template <bool OUT_BF16, int CT, int LPR>
__global__ __launch_bounds__(LB) void edl_l1_bwd_kernel(const float* __restrict__ cls, const long long* __restrict__ labels,
                                                        const float* __restrict__ lw, const float* __restrict__ bp,
                                                        const float* __restrict__ bt, const float* __restrict__ bw, const LossLevels lv, int C,
                                                        float gamma, float alpha, const float* __restrict__ g_cls,
                                                        const float* __restrict__ g_bbox, const float* __restrict__ g_noR, float g_noR_s, int g_noR_bcast,
                                                        void* __restrict__ grad_cls, void* __restrict__ grad_bbox, int A, int pitch_cls,
                                                        int pitch_box, int g_lstride, const float* __restrict__ g_div, int nlv) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  constexpr int RB = LB / LPR;
  const int P = C | 1;
  int level, blk0; long long row0, rend;
  level_of_block(lv, (int)blockIdx.x, level, blk0, row0, rend);
  const long long r0 = row0 + (long long)((int)blockIdx.x - blk0) * RB;
  const int nr = (int)min((long long)RB, rend - r0);
  const int gl = level * g_lstride;          // the level's upstream gradients (per-level scalars of a level-fused launch; 0 otherwise)
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
    // (g_div: the upstream gradients are those of the DIVIDED sums -- the quotient is what autograd's division backward forms)
    const float gc_ = g_div ? g_cls[gl] / g_div[gl] : g_cls[gl];
    const float gn_ = g_noR ? (g_noR_bcast ? (g_div ? g_noR[gl] / g_div[2 * nlv + gl] : g_noR[gl]) : g_noR[r]) : g_noR_s;
    const float coef = gc_ * lw[r] + gn_;
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
      const float gb = g_div ? g_bbox[gl] / g_div[nlv + gl] : g_bbox[gl];
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
  LossLevels lv; long long tot;
  const long long nb = fill_levels(lv, 1, &nrows, LB / edl_lpr(C), tot);
  const int g_lstride = 0, nlv_ = 1;
  const float* const g_div = nullptr;
#define AOD_EDL_BWD(BF, LPR_)                                                                                                              \
  hipLaunchKernelGGL((edl_l1_bwd_kernel<BF, 24, LPR_>), dim3((unsigned)nb), dim3(LB), (size_t)(LB / LPR_) * (C | 1) * 4, (hipStream_t)stream, cls, \
                     (const long long*)labels, label_w, bbox_pred, bbox_tgt, bbox_w, lv, C, gamma, alpha, g_cls, g_bbox, g_noR, \
                     g_noR_scalar, g_noR_is_scalar, grad_cls, grad_bbox, A, pitch_cls, pitch_box, g_lstride, g_div, nlv_)
  if (out_bf16) { if (C <= 24) AOD_EDL_BWD(true, 1); else AOD_EDL_BWD(true, 4); }
  else { if (C <= 24) AOD_EDL_BWD(false, 1); else AOD_EDL_BWD(false, 4); }
  AOD_LAUNCH_CHECK();
  return 0;
}

// g_sums[3][nlevels] = the gradients of aod_edl_focal_l1_levels_fwd's sums (row 0: classification sums, row 1: box sums, row 2: row sums) --
// of the DIVIDED sums when `divisors` (the forward's) is given: the kernel forms g / divisor itself;
// g_noR_rows (optional): a gradient per anchor row of loss_noR, replaces row 2.
extern "C" int aod_edl_focal_l1_levels_bwd(const float* cls, const int64_t* labels, const float* label_w, const float* bbox_pred,
                                           const float* bbox_tgt, const float* bbox_w, int nlevels, const int64_t* level_rows, int C, float gamma,
                                           float alpha, const float* g_sums, const float* divisors, const float* g_noR_rows, void* grad_cls,
                                           void* grad_bbox, int out_bf16, int A, int pitch_cls, int pitch_box, aod_stream_t stream) {
  AOD_CHECK_ARG(nlevels >= 1 && nlevels <= MAXLV && level_rows, "edl_levels_bwd: 1..8 levels");
  AOD_CHECK_ARG(cls && labels && label_w && g_sums && grad_cls, "edl_levels_bwd: null pointer");
  AOD_CHECK_ARG(C >= 1 && C <= MAXC && A >= 1 && pitch_cls >= A * C, "edl_levels_bwd: bad C/A/pitch");
  AOD_CHECK_ARG(!grad_bbox || (bbox_pred && bbox_tgt && bbox_w && pitch_box >= A * 4), "edl_levels_bwd: bbox args");
  LossLevels lv; long long tot;
  const int nb = fill_levels(lv, nlevels, level_rows, LB / edl_lpr(C), tot);
  if (nb == 0) return 0;
  const float* g_cls = g_sums; const float* g_bbox = g_sums + nlevels;
  const float* g_noR = g_noR_rows ? g_noR_rows : g_sums + 2 * nlevels;
  const float g_noR_scalar = 0.f; const int g_noR_is_scalar = g_noR_rows ? 0 : 1, g_lstride = 1, nlv_ = nlevels;
  const float* const g_div = divisors;
  if (out_bf16) { if (C <= 24) AOD_EDL_BWD(true, 1); else AOD_EDL_BWD(true, 4); }
  else { if (C <= 24) AOD_EDL_BWD(false, 1); else AOD_EDL_BWD(false, 4); }
#undef AOD_EDL_BWD
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- MEH loss
__global__ __launch_bounds__(256) void meh_fwd_kernel(const float* __restrict__ lam, const float* __restrict__ loss, const float* __restrict__ bw4,
                                                      const LossLevels lv, float* __restrict__ partials) {
  float s = 0.f;
  int level, blk0; long long row0, n;
  level_of_block(lv, (int)blockIdx.x, level, blk0, row0, n);
  const long long i = row0 + (long long)((int)blockIdx.x - blk0) * 256 + threadIdx.x;
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
__global__ void meh_bwd_kernel(const float* __restrict__ lam, const float* __restrict__ loss, const float* __restrict__ bw4, const LossLevels lv,
                               const float* __restrict__ g, void* __restrict__ grad, int A, int pitch, int g_lstride) {
  int level, blk0; long long row0, n;
  level_of_block(lv, (int)blockIdx.x, level, blk0, row0, n);
  const long long i = row0 + (long long)((int)blockIdx.x - blk0) * 256 + threadIdx.x;
  if (i < n) {
    const float w = bw4[i * 4];
    const float v = g[level * g_lstride] * 2.f * w * w * (lam[i] + 1e-9f - loss[i]);
    const long long o = (i / A) * pitch + (i % A);
    if (OUT_BF16) ((bf16_t*)grad)[o] = (bf16_t)v; else ((float*)grad)[o] = v;
  }
}
extern "C" int aod_meh_loss_fwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n, float* out_sum, float* partials,
                                aod_stream_t stream) {
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && out_sum && partials, "meh_fwd: null pointer");
  if (n == 0) return 0;
  LossLevels lv; long long tot;
  const long long nb = fill_levels(lv, 1, &n, 256, tot);
  hipLaunchKernelGGL(meh_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, 1, out_sum);
  AOD_LAUNCH_CHECK();
  return 0;
}
// all levels in one launch (Lambda_L2.py:235-241 once per level): out_sums[nlevels]
extern "C" int aod_meh_loss_levels_fwd(const float* lam, const float* loss_noR, const float* bbox_w4, int nlevels, const int64_t* level_rows,
                                       float* out_sums, float* partials, aod_stream_t stream) {
  AOD_CHECK_ARG(nlevels >= 1 && nlevels <= MAXLV && level_rows, "meh_levels_fwd: 1..8 levels");
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && out_sums && partials, "meh_levels_fwd: null pointer");
  LossLevels lv; long long tot;
  const int nb = fill_levels(lv, nlevels, level_rows, 256, tot);
  if (nb) hipLaunchKernelGGL(meh_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, partials);
  hipLaunchKernelGGL(reduce_partials_levels_kernel, dim3(nlevels), dim3(256), 0, (hipStream_t)stream, partials, lv, 1, out_sums, 1, nlevels, (const int*)nullptr,
                     0, (float*)nullptr, (float*)nullptr);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_meh_loss_bwd(const float* lam, const float* loss_noR, const float* bbox_w4, int64_t n, const float* g, void* grad_lam,
                                int out_bf16, int A, int pitch, aod_stream_t stream) {
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && g && grad_lam && A >= 1 && pitch >= A, "meh_bwd: bad args");
  if (n == 0) return 0;
  LossLevels lv; long long tot;
  const long long nb = fill_levels(lv, 1, &n, 256, tot);
  if (out_bf16) hipLaunchKernelGGL((meh_bwd_kernel<true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, g, grad_lam, A, pitch, 0);
  else hipLaunchKernelGGL((meh_bwd_kernel<false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, g, grad_lam, A, pitch, 0);
  AOD_LAUNCH_CHECK();
  return 0;
}
extern "C" int aod_meh_loss_levels_bwd(const float* lam, const float* loss_noR, const float* bbox_w4, int nlevels, const int64_t* level_rows,
                                       const float* g, void* grad_lam, int out_bf16, int A, int pitch, aod_stream_t stream) {
  AOD_CHECK_ARG(nlevels >= 1 && nlevels <= MAXLV && level_rows, "meh_levels_bwd: 1..8 levels");
  AOD_CHECK_ARG(lam && loss_noR && bbox_w4 && g && grad_lam && A >= 1 && pitch >= A, "meh_levels_bwd: bad args");
  LossLevels lv; long long tot;
  const int nb = fill_levels(lv, nlevels, level_rows, 256, tot);
  if (nb == 0) return 0;
  if (out_bf16) hipLaunchKernelGGL((meh_bwd_kernel<true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, g, grad_lam, A, pitch, 1);
  else hipLaunchKernelGGL((meh_bwd_kernel<false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lam, loss_noR, bbox_w4, lv, g, grad_lam, A, pitch, 1);
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
