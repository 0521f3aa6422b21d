// HUA epistemic-uncertainty scoring on gfx950: GetObjectIdx + ComputeObjUnc + AggregateObjScaleUnc
// (mmdet/models/dense_heads/Lambda_L2.py:343-349, 489-537, 597-619; torch._sample_dirichlet) as three launches:
//   H1  one block per image : objects (det score > 0.3), (candidate, object) pairs with IoU > 0.5 and
//       max-score > 0.3 in the reference's nonzero() order, per-level pair ranges and mean(lambda).
//   H2  one 256-thread workgroup per pair: lanes = Monte-Carlo samples (LPP = 1) or quarter samples (LPP = 4, up to 96 Dirichlet
//       columns); every lane draws the gamma variates of its classes (Philox4x32-10 counter RNG -> Box-Muller pair -> two
//       Marsaglia-Tsang candidates per block), normalises, accumulates sample entropy and the running sum of p; shuffles + LDS reduce
//       to (aleatoric, epistemic) of the pair.  Vector-ALU bound: ~100 B in, 8 B out, 10^4 variates per pair.
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
  int* pair_cand; int* pair_obj; int* pair_cls; float* pair_epi; float* pair_ale; int* lvl_pair_start; float* lam_mean; int* nobj;
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
  if (!p.scale_mode) {
    // objects = detections with score > thr, in detection order: thread j takes detection j, ordered compaction by ballot + wave offsets
    // (max_num <= HMAXO = 256: the first four waves)
    const int nd = min(p.num_det[b], p.max_num);
    float d5[5];
    bool keep = false;
    if (t < nd) {
      for (int u = 0; u < 5; ++u) d5[u] = dt[t * 5 + u];
      keep = d5[4] > p.obj_score_thr;
    }
    // (one barrier outside the predicate: every thread of the workgroup reaches it)
    const bool head = t < HMAXO;
    const unsigned long long m = __ballot(keep);
    if (head && (t & 63) == 0) s_warp[t >> 6] = __popcll(m);
    const int within = __popcll(m & ((1ull << (t & 63)) - 1ull));
    __syncthreads();
    if (head) {
      int off = 0;
      for (int k = 0; k < (t >> 6); ++k) off += s_warp[k];
      if (keep) {
        const int o = off + within;
        for (int u = 0; u < 4; ++u) obox[o][u] = d5[u];
        obox[o][4] = (d5[2] - d5[0]) * (d5[3] - d5[1]);
      }
      if (t == 0) { const int no = s_warp[0] + s_warp[1] + s_warp[2] + s_warp[3]; s_no = no; p.nobj[b] = no; }
    }
    __syncthreads();      // (s_warp is reused by the pair scan below)
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
        float ms = 0.f;                   // SSD quirk: max incl. the bg column (0 for RetinaNet)
        const float* srow = sc + (long long)i * (p.C + 1);
        if (p.C + 1 <= 24) {              // all loads of the row in flight at once (a run-time loop pays one round trip per class)
          float sv[24];
#pragma unroll
          for (int c = 0; c < 24; ++c) sv[c] = c <= p.C ? srow[c] : 0.f;
#pragma unroll
          for (int c = 0; c < 24; ++c) ms = fmaxf(ms, sv[c]);
        } else {
          for (int c = 0; c <= p.C; ++c) ms = fmaxf(ms, srow[c]);
        }
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
  {
    // mean(lambda) over the pairs of (image, level) (Lambda_L2.py:513-515): one wave per level, lanes stride over the pairs
    const int wv = t >> 6, ln = t & 63;
    if (wv < p.L) {
      int s = 0, e = 0, acc = 0;
      for (int l = 0; l <= wv; ++l) { s = acc; acc += s_lvl[l + 1]; e = acc; }
      s = min(s, p.max_pairs); e = min(e, p.max_pairs);
      float sum = 0.f;
      if (p.scale_mode) {      // l_scores.mean() over every anchor of the level (:551-553)
        s = p.level_start[wv]; e = p.level_start[wv + 1];
        for (int k = s + ln; k < e; k += 64) sum += p.lam[(long long)b * p.n + k];
      } else {
        for (int k = s + ln; k < e; k += 64) sum += p.lam[(long long)b * p.n + pc[k]];
      }
      sum = wave_sum(sum);
      if (ln == 0) p.lam_mean[b * HMAXL + wv] = e > s ? sum / (float)(e - s) : 0.f;
    }
  }
}

// ---------------------------------------------------------------- Philox4x32-10 + samplers (mirrors oracle/hua.py)
// hardware transcendentals (v_log_f32 = log2, v_exp_f32 = exp2, v_sin/v_cos take REVOLUTIONS, v_rcp / v_sqrt): the sampler is bound by
// vector-ALU issue, and the libm forms (range reduction, denormal fix-ups) tripled its instruction count.  Arguments here are normal
// floats in (0, 1] or O(1); the numpy restatement (oracle/hua.py) uses exact float32 functions and agrees to ~1e-6 per variate.
#define LN2_F 0.693147180559945309f
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * LN2_F; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// One Gamma(alpha, 1) variate: Marsaglia-Tsang (2000) on a = alpha (+1 when alpha < 1), d = a - 1/3, cc = 1/sqrt(9d).
// Attempt t draws ONE Philox block at counter (2t, c1, c2, c3): words -> (u_a, u_b) give the Box-Muller pair
// x1 = r cos(2 pi u_b), x2 = r sin(2 pi u_b); (x1, u_c) is the first candidate and (x2, u_d) the second.  Trying two candidates per
// block matters on a 64-lane wavefront: with one candidate (acceptance ~95.5 %) some lane rejects in 95 % of the rounds and the whole
// wave pays a second Philox block; with two, a third-round is needed in ~12 % of the rounds.
__device__ __forceinline__ float gamma_mt(float d, float cc, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  float out = 0.f;
  for (unsigned t = 0; t < 32u; ++t) {
    unsigned r[4];
    philox4x32_10(2u * t, c1, c2, c3, k0, k1, r);
    const float ua = u01(r[0]), ub = u01(r[1]);
    const float rad = __builtin_amdgcn_sqrtf(-2.f * fast_ln(ua));
    const float x1 = rad * __builtin_amdgcn_cosf(ub), x2 = rad * __builtin_amdgcn_sinf(ub);
    const float v1 = 1.f + cc * x1, v2 = 1.f + cc * x2;
    const float w1 = v1 * v1 * v1, w2 = v2 * v2 * v2;
    const bool ok1 = v1 > 0.f && fast_ln(u01(r[2])) < 0.5f * x1 * x1 + d - d * w1 + d * fast_ln(w1);
    const bool ok2 = v2 > 0.f && fast_ln(u01(r[3])) < 0.5f * x2 * x2 + d - d * w2 + d * fast_ln(w2);
    if (ok1) { out = d * w1; break; }
    if (ok2) { out = d * w2; break; }
  }
  return out;
}

template <int LPP>
__device__ __forceinline__ float part_sum(float v) {     // over the LPP lanes that share one Monte-Carlo sample
  if (LPP >= 2) v += __shfl_xor(v, 1, 64);
  if (LPP >= 4) v += __shfl_xor(v, 2, 64);
  return v;
}
template <int LPP>
__device__ __forceinline__ float slot_sum(float v) {     // over the 64 / LPP lanes that hold the same class part
#pragma unroll
  for (int o = 32; o >= LPP; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One 256-thread workgroup per pair (grid-stride over the pairs of image blockIdx.y); the four waves split the Monte-Carlo samples.
// A lane owns ONE sample (LPP = 1: C <= 24 Dirichlet columns -- VOC's 20 / SSD's 21) or a quarter of one (LPP = 4: COCO's 80 / 81 columns,
// the lane's classes are [part * cpl, (part + 1) * cpl)): per-class state (d, cc, 1/alpha, running sum of p) stays in registers for any class
// count up to 96; the sums over a sample's classes cross LPP lanes by shuffle, the sums over samples cross the wave once per pair.
// EXACT: every lane owns exactly CPL classes (nd == CPL * LPP): no predication in the class loops.
template <int CPL, int LPP, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LPP == 1 && CPL <= 21 ? 4 : 2, LPP == 1 && CPL <= 21 ? 4 : 2))) void hua_sample_kernel(const HuaArgs p) {
  constexpr int SPR = 64 / LPP;                      // samples per wave round
  constexpr bool ALIGNED = LPP == 1 || (EXACT && (CPL % 4) == 0);      // a lane's first class is a multiple of four
  __shared__ float s_red[4][HMAXC + 2];
  __shared__ float s_tot[2];
  // a sample's gamma variates wait in LDS for the sample's total (one column per thread: conflict-free) instead of in CPL registers -- with
  // them the 20-class instance needs 147 VGPRs (three waves per SIMD), without 4 waves fit
  __shared__ float s_g[CPL][256];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int part = lane % LPP, slot = lane / LPP;
  const int nd = p.nd;
  const int cpl = EXACT ? CPL : (nd + LPP - 1) / LPP;
  const int c0 = part * cpl;
  const int n = EXACT ? CPL : max(min(cpl, nd - c0), 0);
  const int np = min(p.pair_count[b], p.max_pairs);
  for (int pi = blockIdx.x; pi < np; pi += gridDim.x) {
    const int cand = p.pair_cand[(long long)b * p.max_pairs + pi], obj = p.pair_obj[(long long)b * p.max_pairs + pi];
    const int lvl = level_of(p, cand);
    const float lamv = p.lam[(long long)b * p.n + cand];
    const float lam_hat = p.lam_mean[b * HMAXL + lvl] / (lamv + 1e-7f) * 25.f;
    const float* sc = p.scores + ((long long)b * p.n + cand) * (p.C + 1);
    const unsigned c2 = (unsigned)p.cand_anchor[(long long)b * p.n + cand];
    const unsigned c3 = (unsigned)p.image_ids[b];
    float dd[CPL], binv[CPL], sum_p[CPL];       // (cc = rsq(9 d) is recomputed per variate: one quarter-rate instruction against 20 registers)
    float best = -1.f;
    int am = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const bool on = EXACT || c < n;
      const float sv = on ? sc[c0 + c] : 0.f;
      if (on && sv > best) { best = sv; am = c0 + c; }            // first maximum (torch.argmax), within this lane's part
      const float alpha = sv * lam_hat;
      const bool boost = alpha < 1.f;
      const float a = boost ? alpha + 1.f : alpha;
      dd[c] = a - (1.0f / 3.0f);
      binv[c] = boost ? fast_rcp(alpha) : 0.f;                     // u^(1/alpha) = exp2(log2(u) * binv); 0 -> factor 1 (no boost)
      sum_p[c] = 0.f;
    }
    if (LPP > 1) {                                                 // argmax over the parts: larger value, ties -> lower class index
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (ob > best || (ob == best && oa < am)) { best = ob; am = oa; }
      }
    }
    if (threadIdx.x == 0) p.pair_cls[(long long)b * p.max_pairs + pi] = am;
    float sum_ent = 0.f;
    for (int r = wave; r * SPR < p.num_samples; r += 4) {
      const int s = r * SPR + slot;
      if (s < p.num_samples) {                                     // uniform over the LPP lanes of a sample
        float tot = 0.f;
        const unsigned cs = ((unsigned)obj << 20) | ((unsigned)s << 7);
        unsigned rb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if (EXACT || c < n) {
            const unsigned gc = (unsigned)(c0 + c);
            // boost uniforms: one Philox block per group of four classes, counter word 0 = 1 (the candidates use even words)
            if (ALIGNED ? (c & 3) == 0 : (c == 0 || (gc & 3u) == 0u)) philox4x32_10(1u, cs | (gc >> 2), c2, c3, p.seed_lo, p.seed_hi, rb);
            float d_ = dd[c];
            asm volatile("" : "+v"(d_));          // (opaque: keeps the recomputation of cc inside the sample loop instead of 20 hoisted registers)
            float gv = gamma_mt(d_, __builtin_amdgcn_rsqf(9.f * d_), cs | gc, c2, c3, p.seed_lo, p.seed_hi);
            gv *= fast_exp2(fast_log2(u01(rb[gc & 3u])) * binv[c]);
            gv = fmaxf(gv, FLT_MIN_F);
            s_g[c][threadIdx.x] = gv;
            tot += gv;
          }
        }
        tot = part_sum<LPP>(tot);
        const float inv = fast_rcp(tot);
        float ent = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if (EXACT || c < n) {
            const float pr = fminf(fmaxf(s_g[c][threadIdx.x] * inv, FLT_MIN_F), ONE_MINUS_EPS);
            ent -= pr * fast_log2(pr);
            sum_p[c] += pr;
          }
        }
        sum_ent += ent;
      }
    }
    // sums over this wave's samples -> LDS (lanes 0 .. LPP-1 hold the totals of their class part)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const float v = slot_sum<LPP>(sum_p[c]);
      if (slot == 0 && (EXACT || c < n)) s_red[wave][c0 + c] = v;
    }
    const float we = wave_sum(sum_ent);
    if (lane == 0) s_red[wave][HMAXC] = we;
    __syncthreads();
    const float invn = 1.f / (float)p.num_samples;
    float term = 0.f;
    if ((int)threadIdx.x < nd) {
      const float avg = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]) * invn;
      term = -avg * logf(avg);
    }
    if (threadIdx.x < 128) {
      term = wave_sum(term);
      if (lane == 0) s_tot[wave] = term;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float total = s_tot[0] + s_tot[1];
      const float ale = (s_red[0][HMAXC] + s_red[1][HMAXC] + s_red[2][HMAXC] + s_red[3][HMAXC]) * LN2_F * invn;
      p.pair_epi[(long long)b * p.max_pairs + pi] = total - ale;
      p.pair_ale[(long long)b * p.max_pairs + pi] = ale;
      if (p.pair_out) {
        float* o = p.pair_out + ((long long)b * p.max_pairs + pi) * 4;
        o[0] = (float)cand; o[1] = (float)obj; o[2] = ale; o[3] = total - ale;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float agg_fold(int mode, float acc, float v, int cnt) {
  if (cnt == 0) return v;
  return mode == 2 ? fmaxf(acc, v) : acc + v;
}
__device__ __forceinline__ float agg_final(int mode, float acc, int cnt) { return mode == 1 ? acc / (float)cnt : acc; }

// H3: one block per image.  Per level the pairs (object, argmax class, epistemic) are staged in LDS; a thread per (object, class) bin
// takes the mean over its pairs IN PAIR ORDER (deterministic), a thread per object then folds the classes in class order, the levels in
// level order, and thread 0 the objects in object order (Lambda_L2.py:526-536, 597-619).
constexpr int HRED_STAGE = 2048;
__global__ __launch_bounds__(1024) void hua_reduce_kernel(const HuaArgs p) {
  extern __shared__ float s_bin[];                 // [no][nd] bin means of the current level, NaN = empty
  __shared__ int s_po[HRED_STAGE], s_pcl[HRED_STAGE];
  __shared__ float s_pe[HRED_STAGE];
  __shared__ float s_lvl[HMAXO];                   // running fold over levels per object
  __shared__ int s_lvln[HMAXO];
  __shared__ unsigned long long s_cls[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const int no = p.nobj[b], nd = p.nd;
  if (t < 2) s_cls[t] = 0ull;
  for (int o = t; o < no; o += 1024) { s_lvl[o] = 0.f; s_lvln[o] = 0; }
  const int* po = p.pair_obj + (long long)b * p.max_pairs;
  const int* pcl = p.pair_cls + (long long)b * p.max_pairs;
  const float* pe = p.pair_epi + (long long)b * p.max_pairs;
  __syncthreads();
  for (int l = 0; l < p.L; ++l) {
    const int s = p.lvl_pair_start[b * (HMAXL + 1) + l], e = p.lvl_pair_start[b * (HMAXL + 1) + l + 1];
    if (e <= s) continue;                          // (block-uniform)
    const int nb = no * nd;
    // running (sum, count) per bin in registers across the staged chunks of this level
    float bs[(HMAXO * HMAXC + 1023) / 1024];
    int bc[(HMAXO * HMAXC + 1023) / 1024];
#pragma unroll
    for (int u = 0; u < (HMAXO * HMAXC + 1023) / 1024; ++u) { bs[u] = 0.f; bc[u] = 0; }
    for (int k0 = s; k0 < e; k0 += HRED_STAGE) {
      const int nk = min(HRED_STAGE, e - k0);
      for (int k = t; k < nk; k += 1024) { s_po[k] = po[k0 + k]; s_pcl[k] = pcl[k0 + k]; s_pe[k] = pe[k0 + k]; }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < (HMAXO * HMAXC + 1023) / 1024; ++u) {
        const int bin = u * 1024 + t;
        if (bin < nb) {
          const int o = bin / nd, c = bin - o * nd;
          for (int k = 0; k < nk; ++k)
            if (s_po[k] == o && s_pcl[k] == c) { bs[u] += s_pe[k]; ++bc[u]; }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < (HMAXO * HMAXC + 1023) / 1024; ++u) {
      const int bin = u * 1024 + t;
      if (bin < nb) s_bin[bin] = bc[u] ? bs[u] / (float)bc[u] : __int_as_float(0x7fc00000);
    }
    __syncthreads();
    for (int o = t; o < no; o += 1024) {
      float acc = 0.f;
      int n = 0;
      for (int c = 0; c < nd; ++c) {
        const float v = s_bin[o * nd + c];
        if (v == v) {
          acc = agg_fold(p.agg_class, acc, v, n);
          ++n;
          atomicOr(&s_cls[c >> 6], 1ull << (c & 63));
        }
      }
      if (n) {
        const float lv = agg_final(p.agg_class, acc, n);
        s_lvl[o] = agg_fold(p.agg_scale, s_lvl[o], lv, s_lvln[o]);
        ++s_lvln[o];
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    float acc = 0.f;
    int n = 0;
    for (int o = 0; o < no; ++o)
      if (s_lvln[o]) { acc = agg_fold(p.agg_obj, acc, agg_final(p.agg_scale, s_lvl[o], s_lvln[o]), n); ++n; }
    float v = n ? agg_final(p.agg_obj, acc, n) : 0.f;
    if (p.clsW) v *= (float)(__popcll(s_cls[0]) + __popcll(s_cls[1]));
    p.unc[b] = v;
  }
}

extern "C" size_t aod_hua_ws_bytes(int B, int max_pairs) {
  return (size_t)B * ((size_t)max_pairs * 20 + (HMAXL + 1) * 4 + HMAXL * 4 + 4) + 64;
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
  p.pair_cls = (int*)w; w += (size_t)B * max_pairs * 4;
  p.pair_epi = (float*)w; w += (size_t)B * max_pairs * 4;
  p.pair_ale = (float*)w; w += (size_t)B * max_pairs * 4;
  p.lvl_pair_start = (int*)w; w += (size_t)B * (HMAXL + 1) * 4;
  p.lam_mean = (float*)w; w += (size_t)B * HMAXL * 4;
  p.nobj = (int*)w;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hua_pairs_kernel, dim3(B), dim3(1024), 0, st, p);
  const int gx = max_pairs < 512 ? max_pairs : 512;
#define AOD_HUA_SAMPLE(CPL_, LPP_, EX_) hipLaunchKernelGGL((hua_sample_kernel<CPL_, LPP_, EX_>), dim3(gx, B), dim3(256), 0, st, p)
  if (p.nd == 20) AOD_HUA_SAMPLE(20, 1, true);
  else if (p.nd == 21) AOD_HUA_SAMPLE(21, 1, true);
  else if (p.nd == 80) AOD_HUA_SAMPLE(20, 4, true);
  else if (p.nd <= 24) AOD_HUA_SAMPLE(24, 1, false);
  else AOD_HUA_SAMPLE(24, 4, false);
#undef AOD_HUA_SAMPLE
  hipLaunchKernelGGL(hua_reduce_kernel, dim3(B), dim3(1024), (size_t)max_num * p.nd * 4, st, p);
  AOD_LAUNCH_CHECK();
  return 0;
}
