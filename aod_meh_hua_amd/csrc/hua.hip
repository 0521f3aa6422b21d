// HUA epistemic-uncertainty scoring on gfx950: GetObjectIdx + ComputeObjUnc + AggregateObjScaleUnc
// (mmdet/models/dense_heads/Lambda_L2.py:343-349, 489-537, 597-619; torch._sample_dirichlet) as three launches:
//   H1  one block per image : objects (det score > 0.3), (candidate, object) pairs with IoU > 0.5 and
//       max-score > 0.3 in the reference's nonzero() order, per-level pair ranges and mean(lambda).
//   H2  one WAVEFRONT per pair: lanes = Monte-Carlo samples (8 rounds x 64 >= 500); every lane draws the
//       20 gamma variates of one Dirichlet sample (Philox4x32-10 counter RNG -> Box-Muller -> Marsaglia-Tsang),
//       normalises, accumulates sample entropy and the running mean of p; wave shuffles reduce to
//       (aleatoric, epistemic) of the pair.  ALU/transcendental bound: ~100 B in, 8 B out, 10^4 variates per pair.
//   H3  one block per image : deterministic (pair-order) means per (object, level, class) bin, then
//       class -> scale -> object aggregation (Sum / Avg / Max each) -> one float per image.
// The RNG stream is keyed by (seed, image id, anchor id, object id, sample, class, attempt): results do not
// depend on batch composition or on how the pool is sharded over GPUs.  The same algorithm is restated in
// numpy in oracle/hua.py (philox_*), which the tests compare against value by value.
#include "common.h"

#define FLT_MIN_F 1.17549435e-38f
#define ONE_MINUS_EPS 0.99999988079071044921875f
constexpr int HMAXC = 96;
constexpr int HMAXL = 8;
constexpr int HMAXO = 256;

struct HuaArgs {
  const float* boxes; const float* scores; const float* lam; const int* cand_anchor;
  const float* dets; const int* num_det; const int* any_fg; const long long* image_ids;
  int B, n, L, C, max_num;
  int level_start[HMAXL + 1];
  float obj_score_thr, obj_iou_thr, fg_thr;
  int num_samples; unsigned seed_lo, seed_hi;
  int agg_class, agg_scale, agg_obj, clsW;
  int nd;           // Dirichlet columns: C (evidence head, zero-padded bg column) or C+1 (SSD softmax incl. background)
  int scale_mode;   // 1: Entropy_ALL / ComputeScaleUnc (Lambda_L2.py:539-569): every foreground anchor is a pair of ONE pseudo object,
                    //    lambda mean over ALL anchors of the (image, level); dets / num_det / boxes are not read
  float* unc; float* pair_out; int max_pairs; int* pair_count;
  // workspace
  int* pair_cand; int* pair_obj; float* pair_epi; float* pair_ale; int* lvl_pair_start; float* lam_mean; int* nobj;
};

__device__ __forceinline__ int level_of(const HuaArgs& p, int cand) {
  int l = 0;
  while (l + 1 < p.L && cand >= p.level_start[l + 1]) ++l;
  return l;
}

// bbox_overlaps(cand, det) (iou2d_calculator.py:212-252): area1 = cand, area2 = det
__device__ __forceinline__ float iou_cd(const float* c, float ac, const float* d, float ad) {
  const float w = fmaxf(fminf(c[2], d[2]) - fmaxf(c[0], d[0]), 0.f);
  const float h = fmaxf(fminf(c[3], d[3]) - fmaxf(c[1], d[1]), 0.f);
  const float ov = w * h;
  const float uni = fmaxf(ac + ad - ov, 1e-6f);
  return ov / uni;
}

__global__ __launch_bounds__(1024) void hua_pairs_kernel(const HuaArgs p) {
  __shared__ float obox[HMAXO][5];
  __shared__ int s_warp[16];
  __shared__ int s_no;
  __shared__ int s_lvl[HMAXL + 1];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* bx = p.boxes + (long long)b * p.n * 4;
  const float* sc = p.scores + (long long)b * p.n * (p.C + 1);
  const float* dt = p.dets + (long long)b * p.max_num * 5;
  if (t == 0 && p.scale_mode) { s_no = 1; p.nobj[b] = 1; }
  if (t == 0 && !p.scale_mode) {
    int no = 0;
    const int nd = min(p.num_det[b], p.max_num);
    for (int j = 0; j < nd; ++j)
      if (dt[j * 5 + 4] > p.obj_score_thr) {
        for (int u = 0; u < 4; ++u) obox[no][u] = dt[j * 5 + u];
        obox[no][4] = (obox[no][2] - obox[no][0]) * (obox[no][3] - obox[no][1]);
        ++no;
      }
    s_no = no;
    p.nobj[b] = no;
  }
  if (t <= p.L) s_lvl[t] = 0;
  __syncthreads();
  const int no = s_no;
  int* pc = p.pair_cand + (long long)b * p.max_pairs;
  int* po = p.pair_obj + (long long)b * p.max_pairs;
  int base = 0;
  for (int c0 = 0; c0 < p.n; c0 += 1024) {
    const int i = c0 + t;
    int cnt = 0;
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
    int lvl = 0;
    if (i < p.n && no > 0) {
      lvl = level_of(p, i);
      bool fg = p.any_fg[lvl * p.B + b] != 0;
      if (fg) {
        float ms = 0.f;
        for (int c = 0; c <= p.C; ++c) ms = fmaxf(ms, sc[(long long)i * (p.C + 1) + c]);   // SSD quirk: max incl. the bg column (0 for RetinaNet)
        fg = ms > p.fg_thr;
      }
      if (fg && p.scale_mode) { mk[0] = 1ull; cnt = 1; }
      else if (fg) {
        float q[4];
        for (int u = 0; u < 4; ++u) q[u] = bx[(long long)i * 4 + u];
        const float aq = (q[2] - q[0]) * (q[3] - q[1]);
        for (int o = 0; o < no; ++o)
          if (iou_cd(q, aq, obox[o], obox[o][4]) > p.obj_iou_thr) {
            mk[o >> 6] |= 1ull << (o & 63);
            ++cnt;
          }
      }
    }
    // exclusive scan of cnt over the block
    const int lane = t & 63, w = t >> 6;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    __syncthreads();
    if (lane == 63) s_warp[w] = inc;
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { const int x = s_warp[k]; if (k < w) wbase += x; tot += x; }
    int pos = base + wbase + inc - cnt;
    if (cnt) {
      atomicAdd(&s_lvl[lvl + 1], cnt);
      for (int o = 0; o < no; ++o) {
        const bool hit = (mk[o >> 6] >> (o & 63)) & 1ull;
        if (hit) { if (pos < p.max_pairs) { pc[pos] = i; po[pos] = o; } ++pos; }
      }
    }
    base += tot;
    __syncthreads();
  }
  if (t == 0) {
    p.pair_count[b] = base;
    int acc = 0;
    for (int l = 0; l <= p.L; ++l) { acc += s_lvl[l]; p.lvl_pair_start[b * (HMAXL + 1) + l] = min(acc, p.max_pairs); }
  }
  __syncthreads();
  if (t < p.L) {
    // mean(lambda) over the pairs of (image, level) in pair order (Lambda_L2.py:513-515)
    int s = 0, e = 0, acc = 0;
    for (int l = 0; l <= t; ++l) { s = acc; acc += s_lvl[l + 1]; e = acc; }
    s = min(s, p.max_pairs); e = min(e, p.max_pairs);
    float sum = 0.f;
    if (p.scale_mode) {      // l_scores.mean() over every anchor of the level (:551-553)
      const int c0 = p.level_start[t], c1 = p.level_start[t + 1];
      for (int k = c0; k < c1; ++k) sum += p.lam[(long long)b * p.n + k];
      p.lam_mean[b * HMAXL + t] = c1 > c0 ? sum / (float)(c1 - c0) : 0.f;
    } else {
      for (int k = s; k < e; ++k) sum += p.lam[(long long)b * p.n + pc[k]];
      p.lam_mean[b * HMAXL + t] = e > s ? sum / (float)(e - s) : 0.f;
    }
  }
}

// ---------------------------------------------------------------- Philox4x32-10 + samplers (mirrors oracle/hua.py)
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* r) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
    const unsigned long long p1 = (unsigned long long)c2 * 0xCD9E8D57ull;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 1.0f) * 5.9604644775390625e-08f; }

__device__ __forceinline__ float gamma_philox(float alpha, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  const bool boost = alpha < 1.f;
  const float a = boost ? alpha + 1.f : alpha;
  const float d = a - (1.0f / 3.0f);
  const float c = 1.f / sqrtf(9.f * d);
  float out = 0.f, ub = 1.f;
  for (unsigned t = 0; t < 64u; ++t) {
    unsigned r[4];
    philox4x32_10(t, c1, c2, c3, k0, k1, r);
    const float ua = u01(r[0]), ubx = u01(r[1]), uc = u01(r[2]);
    if (t == 0) ub = u01(r[3]);
    const float x = sqrtf(-2.f * logf(ua)) * cosf(6.283185307179586f * ubx);
    const float v = 1.f + c * x;
    const float v3 = v * v * v;
    if (v > 0.f && logf(uc) < 0.5f * x * x + d - d * v3 + d * logf(v3)) { out = d * v3; break; }
  }
  if (boost) out = out * expf(logf(ub) / alpha);
  return out;
}

// CT > 0: class count known at compile time (20 for VOC) -> alpha / g / sum_p live in registers
template <int CT>
__global__ __launch_bounds__(256) void hua_sample_kernel(const HuaArgs p) {
  constexpr int CA = CT ? CT : HMAXC;
  const int C = CT ? CT : p.nd;
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int np = min(p.pair_count[b], p.max_pairs);
  for (int pi = blockIdx.x * 4 + wave; pi < np; pi += gridDim.x * 4) {
  const int cand = p.pair_cand[(long long)b * p.max_pairs + pi], obj = p.pair_obj[(long long)b * p.max_pairs + pi];
  const int lvl = level_of(p, cand);
  const float lamv = p.lam[(long long)b * p.n + cand];
  const float lam_hat = p.lam_mean[b * HMAXL + lvl] / (lamv + 1e-7f) * 25.f;
  const float* sc = p.scores + ((long long)b * p.n + cand) * (p.C + 1);
  const unsigned c2 = (unsigned)p.cand_anchor[(long long)b * p.n + cand];
  const unsigned c3 = (unsigned)p.image_ids[b];
  float sum_p[CA];
  float alpha[CA];
#pragma unroll
  for (int c = 0; c < C; ++c) { alpha[c] = sc[c] * lam_hat; sum_p[c] = 0.f; }
  float sum_ent = 0.f;
  for (int s0 = 0; s0 < p.num_samples; s0 += 64) {
    const int s = s0 + lane;
    if (s < p.num_samples) {
      float g[CA];
      float tot = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const unsigned c1 = ((unsigned)obj << 20) | ((unsigned)s << 7) | (unsigned)c;
        g[c] = fmaxf(gamma_philox(alpha[c], c1, c2, c3, p.seed_lo, p.seed_hi), FLT_MIN_F);
        tot += g[c];
      }
      float ent = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float pr = fminf(fmaxf(g[c] / tot, FLT_MIN_F), ONE_MINUS_EPS);
        ent -= pr * logf(pr);
        sum_p[c] += pr;
      }
      sum_ent += ent;
    }
  }
  const float inv = 1.f / (float)p.num_samples;
  float total = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float avg = wave_sum(sum_p[c]) * inv;
    total -= avg * logf(avg);
  }
  const float ale = wave_sum(sum_ent) * inv;
  if (lane == 0) {
    p.pair_epi[(long long)b * p.max_pairs + pi] = total - ale;
    p.pair_ale[(long long)b * p.max_pairs + pi] = ale;
    if (p.pair_out) {
      float* o = p.pair_out + ((long long)b * p.max_pairs + pi) * 4;
      o[0] = (float)cand; o[1] = (float)obj; o[2] = ale; o[3] = total - ale;
    }
  }
  }
}

__device__ __forceinline__ float agg_fold(int mode, float acc, float v, int cnt) {
  if (cnt == 0) return v;
  return mode == 2 ? fmaxf(acc, v) : acc + v;
}
__device__ __forceinline__ float agg_final(int mode, float acc, int cnt) { return mode == 1 ? acc / (float)cnt : acc; }

__global__ __launch_bounds__(1024) void hua_reduce_kernel(const HuaArgs p) {
  __shared__ float s_val[HMAXO][HMAXL];
  __shared__ int s_has[HMAXO][HMAXL];
  __shared__ float s_obj[HMAXO];
  __shared__ int s_objhas[HMAXO];
  __shared__ unsigned long long s_cls[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const int no = p.nobj[b];
  if (t < 2) s_cls[t] = 0ull;
  __syncthreads();
  const int* pc = p.pair_cand + (long long)b * p.max_pairs;
  const int* po = p.pair_obj + (long long)b * p.max_pairs;
  const float* pe = p.pair_epi + (long long)b * p.max_pairs;
  const float* sc = p.scores + (long long)b * p.n * (p.C + 1);
  for (int u = t; u < no * p.L; u += 1024) {
    const int o = u / p.L, l = u - o * p.L;
    const int s = p.lvl_pair_start[b * (HMAXL + 1) + l], e = p.lvl_pair_start[b * (HMAXL + 1) + l + 1];
    float sum[HMAXC];
    int cnt[HMAXC];
    for (int c = 0; c < p.nd; ++c) { sum[c] = 0.f; cnt[c] = 0; }
    for (int k = s; k < e; ++k)
      if (po[k] == o) {
        const float* r = sc + (long long)pc[k] * (p.C + 1);
        int am = 0;
        float mv = r[0];
        for (int c = 1; c < p.nd; ++c) if (r[c] > mv) { mv = r[c]; am = c; }   // first max (torch.argmax on CPU)
        sum[am] += pe[k];
        ++cnt[am];
      }
    float acc = 0.f;
    int n = 0;
    for (int c = 0; c < p.nd; ++c)
      if (cnt[c]) {
        acc = agg_fold(p.agg_class, acc, sum[c] / (float)cnt[c], n);
        ++n;
        atomicOr(&s_cls[c >> 6], 1ull << (c & 63));
      }
    s_val[o][l] = n ? agg_final(p.agg_class, acc, n) : 0.f;
    s_has[o][l] = n;
  }
  __syncthreads();
  for (int o = t; o < no; o += 1024) {
    float acc = 0.f;
    int n = 0;
    for (int l = 0; l < p.L; ++l)
      if (s_has[o][l]) { acc = agg_fold(p.agg_scale, acc, s_val[o][l], n); ++n; }
    s_obj[o] = n ? agg_final(p.agg_scale, acc, n) : 0.f;
    s_objhas[o] = n;
  }
  __syncthreads();
  if (t == 0) {
    float acc = 0.f;
    int n = 0;
    for (int o = 0; o < no; ++o)
      if (s_objhas[o]) { acc = agg_fold(p.agg_obj, acc, s_obj[o], n); ++n; }
    float v = n ? agg_final(p.agg_obj, acc, n) : 0.f;
    if (p.clsW) v *= (float)(__popcll(s_cls[0]) + __popcll(s_cls[1]));
    p.unc[b] = v;
  }
}

extern "C" size_t aod_hua_ws_bytes(int B, int max_pairs) {
  return (size_t)B * ((size_t)max_pairs * 16 + (HMAXL + 1) * 4 + HMAXL * 4 + 4) + 64;
}

extern "C" int aod_hua_score(const float* boxes, const float* scores, const float* lam, const int32_t* cand_anchor, const float* dets,
                             const int32_t* num_det, const int32_t* level_start_host, const int32_t* level_any_fg, const int64_t* image_ids,
                             int B, int n, int L, int C, int max_num, float obj_score_thr, float obj_iou_thr, float fg_thr, int num_samples,
                             uint64_t seed, const int32_t* agg3_host, int clsW, int scale_mode, int dirichlet_cols, float* unc, float* pair_out, int max_pairs,
                             int32_t* pair_count, void* ws, aod_stream_t stream) {
  if (B == 0) return 0;
  AOD_CHECK_ARG(scores && lam && cand_anchor && level_start_host && level_any_fg && image_ids && unc && pair_count && ws, "hua: null pointer");
  AOD_CHECK_ARG(scale_mode || (boxes && dets && num_det), "hua: object mode needs boxes / dets / num_det");
  AOD_CHECK_ARG(L >= 1 && L <= HMAXL && C >= 1 && C <= HMAXC && max_num <= HMAXO && num_samples >= 1 && num_samples <= 8192 && max_pairs >= 1,
                "hua: L<=8, C<=96, max_num<=256, samples<=8192 required");
  HuaArgs p;
  p.boxes = boxes; p.scores = scores; p.lam = lam; p.cand_anchor = cand_anchor; p.dets = dets; p.num_det = num_det; p.any_fg = level_any_fg;
  p.image_ids = (const long long*)image_ids; p.B = B; p.n = n; p.L = L; p.C = C; p.max_num = max_num;
  for (int i = 0; i <= L; ++i) p.level_start[i] = level_start_host[i];
  AOD_CHECK_ARG(p.level_start[0] == 0 && p.level_start[L] == n, "hua: level_start must cover [0, n)");
  p.obj_score_thr = obj_score_thr; p.obj_iou_thr = obj_iou_thr; p.fg_thr = fg_thr; p.num_samples = num_samples;
  p.seed_lo = (unsigned)(seed & 0xffffffffull); p.seed_hi = (unsigned)(seed >> 32);
  p.agg_class = agg3_host ? agg3_host[0] : 0; p.agg_scale = agg3_host ? agg3_host[1] : 2; p.agg_obj = agg3_host ? agg3_host[2] : 0; p.clsW = clsW; p.scale_mode = scale_mode; p.nd = dirichlet_cols > 0 ? dirichlet_cols : C;
  AOD_CHECK_ARG(p.nd == C || p.nd == C + 1, "hua: dirichlet_cols must be C or C+1");
  p.unc = unc; p.pair_out = pair_out; p.max_pairs = max_pairs; p.pair_count = pair_count;
  char* w = (char*)ws;
  p.pair_cand = (int*)w; w += (size_t)B * max_pairs * 4;
  p.pair_obj = (int*)w; w += (size_t)B * max_pairs * 4;
  p.pair_epi = (float*)w; w += (size_t)B * max_pairs * 4;
  p.pair_ale = (float*)w; w += (size_t)B * max_pairs * 4;
  p.lvl_pair_start = (int*)w; w += (size_t)B * (HMAXL + 1) * 4;
  p.lam_mean = (float*)w; w += (size_t)B * HMAXL * 4;
  p.nobj = (int*)w;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hua_pairs_kernel, dim3(B), dim3(1024), 0, st, p);
  const int gx = (max_pairs + 3) / 4 < 256 ? (max_pairs + 3) / 4 : 256;
  if (p.nd == 20) hipLaunchKernelGGL((hua_sample_kernel<20>), dim3(gx, B), dim3(256), 0, st, p);
  else if (p.nd == 21) hipLaunchKernelGGL((hua_sample_kernel<21>), dim3(gx, B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((hua_sample_kernel<0>), dim3(gx, B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(hua_reduce_kernel, dim3(B), dim3(1024), 0, st, p);
  AOD_LAUNCH_CHECK();
  return 0;
}
