// Scoring pass kernels (gfx950): softmax/row-max scan, stable per-level top-k, gather + delta decode,
// class-aware greedy multiclass NMS.  Replaces Lambda_L2.py:264-332 + core/post_processing/bbox_nms.py:7-93
// (+ mmcv batched_nms / nms_cpu semantics).  Index outputs are exact functions of the fp32 scores:
//   top-k   : descending score, ties -> lower anchor index (torch.topk's tie order is unspecified; pinned here)
//   NMS     : descending score, ties -> lower flat (candidate*C + class) index; suppress iff
//             inter / (area_i + area_j - inter) > thr on the class-offset boxes (mmcv nms_cpu form).
// HBM-bound row scans read the fp32 logits once with one thread per anchor row; selection / NMS work is
// per (image[, level]) blocks with LDS histograms and an in-LDS bitonic sort.
#include "common.h"

#define MAXC 96

// ---------------------------------------------------------------- shared row math (bit-identical wherever used)
// alphas = softmax(x); S = sum(alphas) + 1e-20; scores = alphas / (S + 1e-9)   (Lambda_L2.py:269-273, gamma = 1)
// has_bg (SSD, My_L_ssd_head.py:331-345): x holds C logits whose LAST column is background; scores = plain softmax over all C,
// the two maxima are taken over the C-1 foreground columns.
// CT: compile-time bound of C.  The loops are fully unrolled and predicated (same order of operations) so that x[] / s[] stay in registers;
// with run-time trip counts the arrays were demoted to scratch memory (784 B per lane).
template <int CT>
__device__ __forceinline__ void row_scores_bg(const float* __restrict__ x, int C, float* s, float& max_alpha, float& max_score) {
  float m = x[0];
#pragma unroll
  for (int c = 1; c < CT; ++c) if (c < C) m = fmaxf(m, x[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < C) { s[c] = expf(x[c] - m); sum += s[c]; }
  max_alpha = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < C) { s[c] = s[c] / sum; if (c < C - 1) max_alpha = fmaxf(max_alpha, s[c]); }
  max_score = max_alpha;
}
template <int CT>
__device__ __forceinline__ void row_scores(const float* __restrict__ x, int C, float* s, float& max_alpha, float& max_score) {
  float m = x[0];
#pragma unroll
  for (int c = 1; c < CT; ++c) if (c < C) m = fmaxf(m, x[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < C) { s[c] = expf(x[c] - m); sum += s[c]; }
  float S = 0.f;
  max_alpha = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < C) { s[c] = s[c] / sum; S += s[c]; max_alpha = fmaxf(max_alpha, s[c]); }
  const float den = (S + 1e-20f) + 1e-9f;
  max_score = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) if (c < C) { s[c] = s[c] / den; max_score = fmaxf(max_score, s[c]); }
}

// ---------------------------------------------------------------- S1: row max of normalised scores + level gate
template <int CT>
__device__ __forceinline__ void rowmax_block(const float* __restrict__ cls, long long rows_per_img, int b, int bx, int C, float fg_thr,
                                             float* __restrict__ rowmax, int* __restrict__ any_fg, int has_bg, float* srow) {
  const int P = C | 1;
  const long long r0 = (long long)bx * 256;
  const int nr = (int)min((long long)256, rows_per_img - r0);
  const float* src = cls + ((long long)b * rows_per_img + r0) * C;
  aod_stage_rows<256>(src, srow, nr, C, P);
  __syncthreads();
  bool fg = false;
  if ((int)threadIdx.x < nr) {
    float x[CT], s[CT], ma, ms;
#pragma unroll
    for (int c = 0; c < CT; ++c) x[c] = c < C ? srow[threadIdx.x * P + c] : 0.f;
    if (has_bg) row_scores_bg<CT>(x, C, s, ma, ms); else row_scores<CT>(x, C, s, ma, ms);
    rowmax[(long long)b * rows_per_img + r0 + threadIdx.x] = ms;
    fg = ma > fg_thr;
  }
  if (__ballot(fg) && (threadIdx.x & 63) == 0) atomicOr(any_fg + b, 1);
}
template <int CT>
__global__ __launch_bounds__(256) void softmax_rowmax_kernel(const float* __restrict__ cls, long long rows_per_img, int B, int C,
                                                             float fg_thr, float* __restrict__ rowmax, int* __restrict__ any_fg, int has_bg) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  rowmax_block<CT>(cls, rows_per_img, blockIdx.y, blockIdx.x, C, fg_thr, rowmax, any_fg, has_bg, srow);
}

extern "C" int aod_softmax_rowmax(const float* cls, int B, int64_t rows_per_img, int C, float fg_thr, float* rowmax, int32_t* any_fg,
                                  int has_bg, aod_stream_t stream) {
  if (B == 0 || rows_per_img == 0) return 0;
  AOD_CHECK_ARG(cls && rowmax && any_fg && C >= 1 && C <= MAXC, "softmax_rowmax: bad args");
  dim3 grid((unsigned)((rows_per_img + 255) / 256), B);
  if (C <= 24)
    hipLaunchKernelGGL(softmax_rowmax_kernel<24>, grid, dim3(256), (size_t)256 * (C | 1) * 4, (hipStream_t)stream, cls, (long long)rows_per_img, B, C,
                       fg_thr, rowmax, any_fg, has_bg);
  else
    hipLaunchKernelGGL(softmax_rowmax_kernel<MAXC>, grid, dim3(256), (size_t)256 * (C | 1) * 4, (hipStream_t)stream, cls, (long long)rows_per_img, B, C,
                       fg_thr, rowmax, any_fg, has_bg);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- block-wide helpers (1024 threads)
constexpr int TB = 1024;

__device__ __forceinline__ int block_excl_scan(int v, int* s_warp, int& total) {
  // exclusive prefix sum over 1024 threads; s_warp: 16 ints of LDS
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  __syncthreads();
  if (lane == 63) s_warp[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < TB / 64; ++i) { const int x = s_warp[i]; if (i < w) base += x; tot += x; }
  total = tot;
  return base + inc - v;
}

// bitonic sort of n (power of two, 64 <= n <= TB) u64 keys in LDS, DESCENDING.  One key per thread, kept in a register: the 45 of 55
// compare-exchange steps (n = 1024) whose partner sits in the same wave are two shuffles, only the 10 with stride >= 64 go through LDS
// (a block-wide barrier costs about as much as everything else in a step).
__device__ void bitonic_desc(unsigned long long* k, int n) {
  __syncthreads();
  const int t = threadIdx.x;
  unsigned long long v = t < n ? k[t] : 0ull;
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      unsigned long long o;
      if (stride >= 64) {
        __syncthreads();                       // earlier partner reads are done
        if (t < n) k[t] = v;
        __syncthreads();
        o = t < n ? k[t ^ stride] : 0ull;
      } else {
        const unsigned lo = __shfl_xor((unsigned)v, stride, 64), hi = __shfl_xor((unsigned)(v >> 32), stride, 64);
        o = ((unsigned long long)hi << 32) | lo;
      }
      const bool keep_max = (((t & stride) == 0) == ((t & size) == 0));     // lower index of a descending run keeps the larger key
      v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
    }
  __syncthreads();
  if (t < n) k[t] = v;
  __syncthreads();
}

// Radix-select over u64 keys produced on the fly by `key(i)` for i in [0, n): finds the value of the k-th
// largest key among those < upper (k >= 1, at least k such keys must exist).  8 passes x 8 bits.
// Histogram increment with wave-level aggregation: score keys share their leading bytes, so a plain atomicAdd per lane serialises a
// whole wave (often the whole block) on one LDS word.  Two rounds of "leader's bin -> ballot -> one add of the popcount" absorb the
// dominant bins; whatever is left (high-entropy digits, little contention) takes the ordinary atomic.
__device__ __forceinline__ void hist_add(int* hist, int bin, bool active) {
  unsigned long long rem = __ballot(active);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (!rem) break;                                   // wave-uniform
    const int leader = __ffsll((long long)rem) - 1;
    const int lb = __shfl(bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == lb);
    if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    rem &= ~same;
    if (bin == lb) active = false;
  }
  if (active) atomicAdd(&hist[bin], 1);
}

template <typename KeyFn>
__device__ unsigned long long radix_select_kth(KeyFn key, long long n, unsigned long long upper, int k, int* hist /*264 LDS ints*/) {
  unsigned long long prefix = 0ull, mask = 0ull;
  int remaining = k;
  for (int pass = 7; pass >= 0; --pass) {
    for (int i = threadIdx.x; i < 256; i += TB) hist[i] = 0;
    __syncthreads();
    const int shift = pass * 8;
    for (long long i0 = 0; i0 < n; i0 += TB) {         // whole waves stay in the loop: hist_add uses wave-wide ballots
      const long long i = i0 + threadIdx.x;
      unsigned long long v = 0ull;
      bool act = false;
      if (i < n) { v = key(i); act = v < upper && (v & mask) == prefix; }
      hist_add(hist, (int)((v >> shift) & 0xffull), act);
    }
    __syncthreads();
    // digit of the k-th key: the bin d with  sum(hist[d+1..255]) < remaining <= sum(hist[d..255]).  Suffix sums by shuffles inside the
    // four waves that hold the 256 bins + their totals through LDS (a serial walk of the histogram is a chain of 256 dependent LDS
    // reads per pass and used to be most of the kernel's time)
    const int lane_ = threadIdx.x & 63, w_ = threadIdx.x >> 6;
    const int h_ = threadIdx.x < 256 ? hist[threadIdx.x] : 0;
    int suf = h_;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_down(suf, o, 64);
      if (lane_ + o < 64) suf += v;
    }
    if (threadIdx.x < 256 && lane_ == 0) hist[256 + w_] = suf;
    __syncthreads();
    if (threadIdx.x < 256) {
      int above = 0;
      for (int ww = w_ + 1; ww < 4; ++ww) above += hist[256 + ww];
      const int incl = suf + above, excl = incl - h_;
      if (excl < remaining && remaining <= incl) { hist[260] = threadIdx.x; hist[261] = excl; hist[263] = (h_ == remaining - excl) ? 1 : 0; }
    }
    __syncthreads();
    const int d = hist[260], acc = hist[261];
    const bool whole_bin = hist[263] != 0;
    remaining -= acc;
    prefix |= ((unsigned long long)d) << shift;
    mask |= 0xffull << shift;
    __syncthreads();
    // every key left in the selected bin is needed (always the case once the score bits are resolved and there are no ties): the
    // threshold `key >= prefix` with the undetermined low digits at zero selects exactly the same set, the remaining passes are moot
    if (whole_bin) break;
  }
  return prefix;
}

// ---------------------------------------------------------------- S2: stable top-k per (image, level)
// key = (score bits << 32) | (0xffffffff - index): larger key = higher score, then lower index; keys are unique.
struct TopkLds {
  int hist[264];
  int s_warp[TB / 64];
  unsigned long long keys[1024];
};
// top-k of one row by the whole block; idx_out[0..k) = anchor indices in descending (score, then lower index) order
__device__ void topk_block(const float* __restrict__ s, long long A, int k, int* __restrict__ idx_out, int cache_n, TopkLds& L, float* s_cache) {
  int* const hist = L.hist;
  unsigned long long* const keys = L.keys;
  const long long ncache = A < (long long)cache_n ? A : (long long)cache_n;
  {
    // the row into LDS as 16-B pieces, up to twelve per thread requested before the first one is stored (one 4-B load per loop trip made
    // the fill of a 36 864-anchor level a chain of 36 memory round trips)
    const bool al = ((((size_t)s) | ((size_t)s_cache)) & 15) == 0;
    const long long n4 = al ? ncache / 4 : 0;
    for (long long i0 = threadIdx.x; i0 < n4; i0 += 12 * TB) {
      f32x4 v[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) if (i0 + k * TB < n4) v[k] = reinterpret_cast<const f32x4*>(s)[i0 + k * TB];
#pragma unroll
      for (int k = 0; k < 12; ++k) if (i0 + k * TB < n4) reinterpret_cast<f32x4*>(s_cache)[i0 + k * TB] = v[k];
    }
    for (long long i = n4 * 4 + threadIdx.x; i < ncache; i += TB) s_cache[i] = s[i];
  }
  __syncthreads();
  auto key = [&](long long i) {
    const float v = i < ncache ? s_cache[i] : s[i];
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
  };
  const unsigned long long kth = radix_select_kth(key, A, ~0ull, k, hist);
  // keys >= kth (exactly k of them) into LDS in ANY order -- the sort follows: one wave-aggregated counter instead of a block scan
  for (int i = threadIdx.x; i < 1024; i += TB) keys[i] = 0ull;
  if (threadIdx.x == 0) hist[262] = 0;
  __syncthreads();
  for (long long c0 = 0; c0 < A; c0 += TB) {
    const long long i = c0 + threadIdx.x;
    unsigned long long kv = 0ull;
    bool take = false;
    if (i < A) { kv = key(i); take = kv >= kth; }
    const unsigned long long m = __ballot(take);
    if (m) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&hist[262], __popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      if (take) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = kv;
    }
  }
  bitonic_desc(keys, 1024);
  for (int i = threadIdx.x; i < k; i += TB) idx_out[i] = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
}
__global__ __launch_bounds__(TB) void topk_kernel(const float* __restrict__ score, long long A, int k, int* __restrict__ idx_out, long long out_pitch,
                                                  int cache_n) {
  __shared__ TopkLds L;
  extern __shared__ float s_cache[];          // the row's scores: the 8 select passes + the compaction re-read them, and a dependent
  const int b = blockIdx.x;                   // L2 round trip per 1024 elements and pass is what the kernel's time used to be
  topk_block(score + (long long)b * A, A, k, idx_out + (long long)b * out_pitch, cache_n, L, s_cache);
}

extern "C" int aod_topk_stable(const float* score, int B, int64_t A, int k, int32_t* idx, int64_t out_pitch, aod_stream_t stream) {
  if (B == 0 || k == 0) return 0;
  AOD_CHECK_ARG(score && idx && k >= 1 && k <= 1024 && k <= A && out_pitch >= k, "topk: need 1 <= k <= min(1024, A)");
  // static LDS of the kernel: keys 8 KB + histogram 1 KB + scan; the rest of the 160 KB caches the row
  const int cache_n = (int)(A < 36864 ? A : 36864);
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4);
  }
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(TB), (size_t)cache_n * 4, (hipStream_t)stream, score, (long long)A, k, idx, (long long)out_pitch, cache_n);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- S3: gather + decode one level
struct GatherArgs {
  const float* cls; const float* reg; const float* lam_map; const float* anchors; const int* idx;
  int B; long long A; int k; int C; long long idx_pitch;
  const float* img_hw; const float* scale4;
  float means[4], stds[4]; float max_ratio;
  float* boxes; float* scores; float* lam; int* cand_anchor;
  long long n_total; long long cand0; long long anchor0; int normalize;
};
template <int CT>
__device__ __forceinline__ void gather_one(const GatherArgs& p, int b, int j, long long a) {
  const long long row = (long long)b * p.A + a;
  float x[CT], s[CT], ma, ms;
#pragma unroll
  for (int c = 0; c < CT; ++c) x[c] = c < p.C ? p.cls[row * p.C + c] : 0.f;
  if (p.normalize == 2) row_scores_bg<CT>(x, p.C, s, ma, ms); else row_scores<CT>(x, p.C, s, ma, ms);
  if (p.normalize == 0) {   // raw softmax: undo nothing, recompute without the (S + 1e-20 + 1e-9) division
    float m = x[0];
#pragma unroll
    for (int c = 1; c < CT; ++c) if (c < p.C) m = fmaxf(m, x[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < p.C) { s[c] = expf(x[c] - m); sum += s[c]; }
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < p.C) s[c] = s[c] / sum;
  }
  const long long o = (long long)b * p.n_total + p.cand0 + j;
  if (p.normalize == 2) {
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < p.C) p.scores[o * p.C + c] = s[c];          // C columns, the last one is the background probability
  } else {
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < p.C) p.scores[o * (p.C + 1) + c] = s[c];
    p.scores[o * (p.C + 1) + p.C] = 0.f;
  }
  p.lam[o] = p.lam_map[row];
  p.cand_anchor[o] = (int)(p.anchor0 + a);
  // delta2bbox (delta_xywh_bbox_coder.py:144-262)
  const f32x4 an = *reinterpret_cast<const f32x4*>(p.anchors + a * 4);
  const f32x4 d0 = *reinterpret_cast<const f32x4*>(p.reg + row * 4);
  float d[4];
  for (int i = 0; i < 4; ++i) d[i] = d0[i] * p.stds[i] + p.means[i];
  const float px = (an[0] + an[2]) * 0.5f, py = (an[1] + an[3]) * 0.5f, pw = an[2] - an[0], ph = an[3] - an[1];
  const float dxw = pw * d[0], dyh = ph * d[1];
  const float dw = fminf(fmaxf(d[2], -p.max_ratio), p.max_ratio), dh = fminf(fmaxf(d[3], -p.max_ratio), p.max_ratio);
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + dxw, gy = py + dyh;
  float bx[4] = {gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f};
  const float Hh = p.img_hw[b * 2], Ww = p.img_hw[b * 2 + 1];
  const float mx[4] = {Ww, Hh, Ww, Hh};
  for (int i = 0; i < 4; ++i) {
    float v = bx[i];
    v = v < 0.f ? 0.f : v;
    v = v > mx[i] ? mx[i] : v;
    if (p.scale4) v = v / p.scale4[b * 4 + i];
    p.boxes[o * 4 + i] = v;
  }
}
template <int CT>
__global__ __launch_bounds__(256) void gather_decode_kernel(const GatherArgs p) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.k) return;
  const long long a = p.idx ? p.idx[(long long)b * p.idx_pitch + j] : j;
  gather_one<CT>(p, b, j, a);
}

extern "C" int aod_gather_decode(const float* cls, const float* reg, const float* lam_map, const float* anchors, const int32_t* idx,
                                 int B, int64_t A, int k, int C, int64_t idx_pitch, const float* img_hw, const float* scale4,
                                 const float* means4, const float* stds4, float wh_ratio_clip, float* boxes, float* scores, float* lam,
                                 int32_t* cand_anchor, int64_t n_total, int64_t cand0, int64_t anchor0, int normalize, aod_stream_t stream) {
  if (B == 0 || k == 0) return 0;
  AOD_CHECK_ARG(cls && reg && lam_map && anchors && img_hw && boxes && scores && lam && cand_anchor && C <= MAXC, "gather_decode: bad args");
  GatherArgs p;
  p.cls = cls; p.reg = reg; p.lam_map = lam_map; p.anchors = anchors; p.idx = idx; p.B = B; p.A = A; p.k = k; p.C = C; p.idx_pitch = idx_pitch;
  p.img_hw = img_hw; p.scale4 = scale4;
  for (int i = 0; i < 4; ++i) { p.means[i] = means4 ? means4[i] : 0.f; p.stds[i] = stds4 ? stds4[i] : 1.f; }
  p.max_ratio = fabsf(logf(wh_ratio_clip));
  p.boxes = boxes; p.scores = scores; p.lam = lam; p.cand_anchor = cand_anchor; p.n_total = n_total; p.cand0 = cand0; p.anchor0 = anchor0; p.normalize = normalize;
  if (C <= 24) hipLaunchKernelGGL(gather_decode_kernel<24>, dim3((k + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gather_decode_kernel<MAXC>, dim3((k + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, p);
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- S1-S3 of ALL pyramid levels in two launches
// As 13 launches (5 x row max, 3 x top-k, 5 x gather) the pre-NMS stage is a chain of 6-80 us kernels that occupy 16-150 CUs each: 330 us
// per 16-image batch of which the level-0 top-k alone is 78.  The level chains are independent: (1) one launch scans every level's logits,
// (2) one launch runs a workgroup per (image, level) that selects the level's top-k (where k < A) and gathers + decodes its candidates.
constexpr int MAXL = 8;
struct LevelsArgs {
  const float* cls[MAXL]; const float* reg[MAXL]; const float* lam_map[MAXL]; const float* anchors[MAXL];
  float* rowmax[MAXL]; int* idx[MAXL];                   // idx[l] = null: the level keeps all its anchors (k == A)
  long long A[MAXL]; int k[MAXL]; int blk0[MAXL + 1]; long long cand0[MAXL]; long long anchor0[MAXL];
  int L, B, C, has_bg, normalize, cache_n;
  float fg_thr;
  int* any_fg;                                           // [L][B]
  GatherArgs g;                                          // the level-independent part (outputs, image sizes, coder constants)
};
template <int CT>
__global__ __launch_bounds__(256) void softmax_rowmax_levels_kernel(const LevelsArgs p) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  int l = 0;
  while (l + 1 < p.L && (int)blockIdx.x >= p.blk0[l + 1]) ++l;
  rowmax_block<CT>(p.cls[l], p.A[l], blockIdx.y, blockIdx.x - p.blk0[l], p.C, p.fg_thr, p.rowmax[l], p.any_fg + l * p.B, p.has_bg, srow);
}
template <int CT>
__global__ __launch_bounds__(TB) void topk_gather_levels_kernel(const LevelsArgs p) {
  __shared__ TopkLds Lds;
  extern __shared__ float s_cache[];
  const int b = blockIdx.x, l = blockIdx.y;
  const long long A = p.A[l];
  const int k = p.k[l];
  GatherArgs g = p.g;
  g.cls = p.cls[l]; g.reg = p.reg[l]; g.lam_map = p.lam_map[l]; g.anchors = p.anchors[l]; g.A = A; g.k = k; g.cand0 = p.cand0[l]; g.anchor0 = p.anchor0[l];
  if (p.idx[l]) {
    topk_block(p.rowmax[l] + (long long)b * A, A, k, p.idx[l] + (long long)b * k, p.cache_n, Lds, s_cache);
    for (int j = threadIdx.x; j < k; j += TB)
      gather_one<CT>(g, b, j, (long long)(0xffffffffu - (unsigned)(Lds.keys[j] & 0xffffffffull)));
  } else {
    for (int j = threadIdx.x; j < k; j += TB) gather_one<CT>(g, b, j, (long long)j);
  }
}

extern "C" int aod_pre_nms_levels(int L, const float* const* cls, const float* const* reg, const float* const* lam_map,
                                  const float* const* anchors, const int64_t* A, const int32_t* k, int B, int C, float fg_thr, int has_bg,
                                  int normalize, const float* img_hw, const float* scale4, const float* means4, const float* stds4,
                                  float wh_ratio_clip, float* rowmax, int32_t* any_fg, int32_t* idx, float* boxes, float* scores, float* lam,
                                  int32_t* cand_anchor, int64_t n_total, aod_stream_t stream) {
  if (B == 0 || L == 0) return 0;
  AOD_CHECK_ARG(L >= 1 && L <= MAXL && cls && reg && lam_map && anchors && A && k && C >= 1 && C <= MAXC, "pre_nms_levels: bad args");
  AOD_CHECK_ARG(img_hw && rowmax && any_fg && boxes && scores && lam && cand_anchor, "pre_nms_levels: null pointer");
  LevelsArgs p;
  memset(&p, 0, sizeof(p));
  long long a0 = 0, c0 = 0, i0 = 0, amax = 0;
  int nb = 0;
  for (int l = 0; l < L; ++l) {
    AOD_CHECK_ARG(A[l] >= 1 && k[l] >= 1 && k[l] <= A[l] && k[l] <= 1024, "pre_nms_levels: need 1 <= k <= min(1024, A) per level");
    AOD_CHECK_ARG(k[l] == A[l] || idx, "pre_nms_levels: idx buffer needed for levels with k < A");
    p.cls[l] = cls[l]; p.reg[l] = reg[l]; p.lam_map[l] = lam_map[l]; p.anchors[l] = anchors[l];
    p.A[l] = A[l]; p.k[l] = k[l];
    p.rowmax[l] = rowmax + (long long)B * a0;
    p.idx[l] = k[l] < A[l] ? idx + (long long)B * i0 : nullptr;
    if (k[l] < A[l]) { i0 += k[l]; amax = A[l] > amax ? A[l] : amax; }
    p.blk0[l] = nb; nb += (int)((A[l] + 255) / 256);
    p.cand0[l] = c0; p.anchor0[l] = a0;
    c0 += k[l]; a0 += A[l];
  }
  p.blk0[L] = nb;
  AOD_CHECK_ARG(c0 == n_total, "pre_nms_levels: n_total must be the sum of k");
  p.L = L; p.B = B; p.C = C; p.has_bg = has_bg; p.normalize = normalize; p.fg_thr = fg_thr; p.any_fg = any_fg;
  p.cache_n = (int)(amax < 36864 ? amax : 36864);
  GatherArgs& g = p.g;
  g.B = B; g.C = C; g.idx = nullptr; g.idx_pitch = 0; g.img_hw = img_hw; g.scale4 = scale4;
  for (int i = 0; i < 4; ++i) { g.means[i] = means4 ? means4[i] : 0.f; g.stds[i] = stds4 ? stds4[i] : 1.f; }
  g.max_ratio = fabsf(logf(wh_ratio_clip));
  g.boxes = boxes; g.scores = scores; g.lam = lam; g.cand_anchor = cand_anchor; g.n_total = n_total; g.normalize = normalize;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_gather_levels_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_gather_levels_kernel<MAXC>), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4);
  }
  const size_t srow = (size_t)256 * (C | 1) * 4, cache = (size_t)p.cache_n * 4;
  if (C <= 24) {
    hipLaunchKernelGGL(softmax_rowmax_levels_kernel<24>, dim3(nb, B), dim3(256), srow, (hipStream_t)stream, p);
    hipLaunchKernelGGL(topk_gather_levels_kernel<24>, dim3(B, L), dim3(TB), cache, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(softmax_rowmax_levels_kernel<MAXC>, dim3(nb, B), dim3(256), srow, (hipStream_t)stream, p);
    hipLaunchKernelGGL(topk_gather_levels_kernel<MAXC>, dim3(B, L), dim3(TB), cache, (hipStream_t)stream, p);
  }
  AOD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- S4: multiclass NMS, one block per image
// workspace per image: vflat[n*C] ints (flat index of the p-th valid entry)
#ifdef AOD_TILE_TIMING
__device__ unsigned long long* g_nms_stamps = nullptr;      // debug build (tools/dbg/nms_timing.py): phase stamps of image 0
#define NSTAMP(k) do { if (g_nms_stamps && threadIdx.x == 0 && blockIdx.x == 0) g_nms_stamps[k] = wall_clock64(); } while (0)
__device__ __forceinline__ bool g_nms_stamps_on() { return g_nms_stamps != nullptr; }
__device__ __forceinline__ void g_nms_stamps_extra(int nvalid, int nk) { g_nms_stamps[8] = nvalid; g_nms_stamps[9] = nk; }
extern "C" int aod_dbg_set_nms_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nms_stamps), &buf, sizeof(buf)); }
#else
#define NSTAMP(k) do {} while (0)
__device__ __forceinline__ bool g_nms_stamps_on() { return false; }
__device__ __forceinline__ void g_nms_stamps_extra(int, int) {}
#endif
constexpr int TR = 1024;  // tranche of candidates sorted at a time (max_num <= 256 detections usually come from the first one)
__global__ __launch_bounds__(TB) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n, int C, float score_thr,
                                                 float iou_thr, int max_num, float* __restrict__ dets, long long* __restrict__ det_labels,
                                                 long long* __restrict__ keep, int* __restrict__ num_det, int* __restrict__ ws_vflat, int kcache_n) {
  // score bits of the first kcache_n valid entries, kept while they are compacted (the low word of a sort key is a function of the entry's
  // position): the radix-select passes and the gathers then read LDS instead of two dependent global loads per key (vflat -> score);
  // entries beyond the cache take the global path.  (As 64-bit keys only 12 288 entries fitted and half of the bench pool's 25 k valid
  // entries per image took the global path: radix select 38 -> 25 us.)
  extern __shared__ unsigned scache[];
  __shared__ int hist[264];
  __shared__ int s_warp[TB / 64];
  __shared__ unsigned long long keys[TR];
  __shared__ float cbox[TR][4];      // the tranche's candidates in sorted order: raw box, class (gathered by all threads before the serial scan)
  __shared__ int ccls[TR];
  __shared__ float kbox[256][4];
  __shared__ float karea[256];
  __shared__ int kcls[256];
  __shared__ int s_nkept, s_nvalid;
  __shared__ float s_maxc;
  const int b = blockIdx.x;
  const float* bx = boxes + (long long)b * n * 4;
  const float* sc = scores + (long long)b * n * (C + 1);
  int* vflat = ws_vflat + (long long)b * n * C;
  const long long NC = (long long)n * C;
  NSTAMP(0);
  // 1. compaction of valid (score > thr) entries in flat order + max coordinate of their boxes
  if (threadIdx.x == 0) { s_nkept = 0; s_maxc = -INFINITY; }
  __syncthreads();
  int base = 0;
  float mymax = -INFINITY;
  // one candidate (row of C scores) per thread and iteration: count its valid classes, one block scan per 1024 candidates, then write its
  // entries in class order -- flat (candidate, class) order as before, 4 scans instead of 10 for 3 765 x 20, and the box of a candidate
  // is read once for the coordinate maximum instead of once per valid class (the scores of the second pass are cache hits)
  for (int c0 = 0; c0 < n; c0 += TB) {
    const int cand = c0 + (int)threadIdx.x;
    int cnt = 0;
    const float* srow = sc + (long long)cand * (C + 1);
    unsigned vm = 0;                 // C <= 24: the row stays in registers -- all its loads in flight at once (a run-time class loop issues one
    float sv[24];                    // load per round trip), valid classes as a bit mask
    if (cand < n) {
      if (C <= 24) {
#pragma unroll
        for (int cl = 0; cl < 24; ++cl) sv[cl] = cl < C ? srow[cl] : -INFINITY;
#pragma unroll
        for (int cl = 0; cl < 24; ++cl) vm |= (sv[cl] > score_thr ? 1u : 0u) << cl;
        cnt = __popc(vm);
      } else {
        for (int cl = 0; cl < C; ++cl) cnt += srow[cl] > score_thr ? 1 : 0;
      }
      if (cnt) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(bx + (long long)cand * 4);
        mymax = fmaxf(mymax, fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])));
      }
    }
    int tot;
    int pos = base + block_excl_scan(cnt, s_warp, tot);
    if (cnt) {
      if (C <= 24) {
#pragma unroll
        for (int cl = 0; cl < 24; ++cl)
          if ((vm >> cl) & 1u) {
            vflat[pos] = cand * C + cl;
            if (pos < kcache_n) scache[pos] = __float_as_uint(sv[cl]);
            ++pos;
          }
      } else {
        for (int cl = 0; cl < C; ++cl) {
          const float v = srow[cl];
          if (v > score_thr) {
            vflat[pos] = cand * C + cl;
            if (pos < kcache_n) scache[pos] = __float_as_uint(v);
            ++pos;
          }
        }
      }
    }
    base += tot;
    __syncthreads();
  }
  const int nvalid = base;
  NSTAMP(1);
  mymax = wave_max(mymax);
  if ((threadIdx.x & 63) == 0 && mymax > -INFINITY) atomicMax((int*)&s_maxc, __float_as_int(mymax));  // coords >= 0 after clipping
  __syncthreads();
  // boxes are clipped to >= 0 so the int ordering of the float bits is valid; guard the (never expected) negative case
  const float off_unit = s_maxc + 1.f;
  auto key = [&](long long pidx) {
    if (pidx < kcache_n) return ((unsigned long long)scache[pidx] << 32) | (unsigned long long)(0xffffffffu - (unsigned)pidx);
    const int f = vflat[pidx];
    const int cand = f / C, cl = f - cand * C;
    return ((unsigned long long)__float_as_uint(sc[(long long)cand * (C + 1) + cl]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)pidx);
  };
  unsigned long long upper = ~0ull;
  int done = 0;
  while (done < nvalid) {
    if (s_nkept >= max_num) break;
    const int take = min(TR, nvalid - done);
    const unsigned long long kth = radix_select_kth(key, nvalid, upper, take, hist);
    NSTAMP(2);
    int np2 = 64;
    while (np2 < take) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += TB) keys[i] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) hist[262] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nvalid; c0 += TB) {          // any order: the sort follows
      const int i = c0 + threadIdx.x;
      unsigned long long kv = 0ull;
      bool t = false;
      if (i < nvalid) { kv = key(i); t = kv < upper && kv >= kth; }
      const unsigned long long m = __ballot(t);
      if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        int b0 = 0;
        if (lane == leader) b0 = atomicAdd(&hist[262], __popcll(m));
        b0 = __shfl(b0, leader, 64);
        if (t) keys[b0 + __popcll(m & ((1ull << lane) - 1ull))] = kv;
      }
    }
    bitonic_desc(keys, np2);
    for (int i = threadIdx.x; i < take; i += TB) {
      const int pidx = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
      const int f = vflat[pidx];
      const int cand = f / C;
      const f32x4 r = *reinterpret_cast<const f32x4*>(bx + (long long)cand * 4);
      cbox[i][0] = r[0]; cbox[i][1] = r[1]; cbox[i][2] = r[2]; cbox[i][3] = r[3];
      ccls[i] = f - cand * C;
    }
    __syncthreads();
    NSTAMP(3);
    // greedy scan by wave 0, 64 candidates per round
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      int nk = s_nkept;      // wave-uniform register copy; LDS is only written back after the scan
      for (int c0 = 0; c0 < take; c0 += 64) {
        if (nk >= max_num) break;
        const int i = c0 + lane;
        const bool act = i < take;
        float q[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0}, area = 0.f, scv = 0.f;
        int cl = -1, pidx = 0;
        if (act) {
          const unsigned long long kv = keys[i];
          pidx = (int)(0xffffffffu - (unsigned)(kv & 0xffffffffull));
          scv = __uint_as_float((unsigned)(kv >> 32));
          cl = ccls[i];
          const float off = (float)cl * off_unit;
          for (int u = 0; u < 4; ++u) { r[u] = cbox[i][u]; q[u] = r[u] + off; }
          area = (q[2] - q[0]) * (q[3] - q[1]);
        }
        bool sup = !act;
        for (int j = 0; j < nk && !sup; ++j) {
          if (kcls[j] != cl) continue;
          const float w = fmaxf(0.f, fminf(q[2], kbox[j][2]) - fmaxf(q[0], kbox[j][0]));
          const float h = fmaxf(0.f, fminf(q[3], kbox[j][3]) - fmaxf(q[1], kbox[j][1]));
          const float inter = w * h;
          const float ovr = inter / (karea[j] + area - inter);
          if (ovr > iou_thr) sup = true;
        }
        // intra-round resolution in order
        for (int s = 0; s < 64; ++s) {
          const bool s_keep = __shfl((int)(!sup), s, 64) != 0;
          if (!s_keep) continue;                    // uniform
          if (nk >= max_num) break;                 // uniform
          const float b0 = __shfl(q[0], s, 64), b1 = __shfl(q[1], s, 64), b2 = __shfl(q[2], s, 64), b3 = __shfl(q[3], s, 64);
          const float ar = __shfl(area, s, 64);
          const int cs = __shfl(cl, s, 64);
          if (lane == s) {
            kbox[nk][0] = q[0]; kbox[nk][1] = q[1]; kbox[nk][2] = q[2]; kbox[nk][3] = q[3]; karea[nk] = area; kcls[nk] = cl;
            float* o = dets + ((long long)b * max_num + nk) * 5;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = scv;
            det_labels[(long long)b * max_num + nk] = cl;
            keep[(long long)b * max_num + nk] = pidx;
          }
          ++nk;
          if (lane > s && !sup && cl == cs) {
            const float w = fmaxf(0.f, fminf(q[2], b2) - fmaxf(q[0], b0));
            const float h = fmaxf(0.f, fminf(q[3], b3) - fmaxf(q[1], b1));
            const float inter = w * h;
            const float ovr = inter / (ar + area - inter);
            if (ovr > iou_thr) sup = true;
          }
        }
      }
      if (lane == 0) s_nkept = nk;
    }
    __syncthreads();
    NSTAMP(4);
    upper = kth;
    done += take;
  }
  __syncthreads();
  if (threadIdx.x == 0) { num_det[b] = s_nkept; }
  // zero the unused tail so that consumers can read fixed-size tensors
  const int nk = s_nkept;
  for (int i = nk * 5 + threadIdx.x; i < max_num * 5; i += TB) dets[(long long)b * max_num * 5 + i] = 0.f;
  for (int i = nk + threadIdx.x; i < max_num; i += TB) { det_labels[(long long)b * max_num + i] = -1; keep[(long long)b * max_num + i] = -1; }
  if (threadIdx.x == 0) s_nvalid = nvalid;
  NSTAMP(5);
  if (g_nms_stamps_on()) { if (threadIdx.x == 0 && blockIdx.x == 0) g_nms_stamps_extra(nvalid, nk); }
}

extern "C" size_t aod_nms_ws_bytes(int B, int n, int C) { return (size_t)B * n * C * 4; }

extern "C" int aod_multiclass_nms(const float* boxes, const float* scores, int B, int n, int C, float score_thr, float iou_thr, int max_num,
                                  float* dets, int64_t* det_labels, int64_t* keep, int32_t* num_det, void* ws, aod_stream_t stream) {
  if (B == 0) return 0;
  AOD_CHECK_ARG(boxes && scores && dets && det_labels && keep && num_det && ws, "nms: null pointer");
  AOD_CHECK_ARG(max_num >= 1 && max_num <= 256 && C >= 1, "nms: max_num must be in 1..256");
  // score cache: up to 28 672 valid (candidate, class) entries per image (112 KB of LDS beside the 36 KB of static arrays)
  const long long nc = (long long)n * C;
  const int kcache_n = (int)(nc < 28672 ? nc : 28672);
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 28672 * 4);
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(TB), (size_t)kcache_n * 4, (hipStream_t)stream, boxes, scores, n, C, score_thr, iou_thr, max_num, dets,
                     (long long*)det_labels, (long long*)keep, num_det, (int*)ws, kcache_n);
  AOD_LAUNCH_CHECK();
  return 0;
}
