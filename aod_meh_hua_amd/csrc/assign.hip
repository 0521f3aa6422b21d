// MaxIoU target assignment + PseudoSampler + delta encoding for a whole batch in three launches.
// Bit-exact w.r.t. the reference's fp32 torch ops (compiled with -ffp-contract=off; every
// intermediate is an explicit fp32 op in the reference's order):
//   bbox_overlaps        mmdet/core/bbox/iou_calculators/iou2d_calculator.py:212-252
//   assign_wrt_overlaps  mmdet/core/bbox/assigners/max_iou_assigner.py:127-210
//   bbox2delta           mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:98-140
//   target assembly      mmdet/models/dense_heads/L_anchor_head.py:155-202 (+ unmap fill values)
// One thread per (image, anchor); the image's gt boxes sit in LDS.  Pass 1: per-anchor max/argmax
// over gts and per-gt max over anchors (atomicMax on the IoU bit pattern, IoU >= 0).  Pass 2
// (only for gt_max_assign_all == 0): first anchor attaining each gt's max.  Pass 3: assignment
// rules in the reference's order (later gt wins ties), labels, weights, encoded targets, #pos.
#include "common.h"

constexpr int GMAX = 128;

__device__ __forceinline__ float iou_ref(float ax1, float ay1, float ax2, float ay2, float area_a, float bx1, float by1, float bx2,
                                         float by2, float area_b) {
  // bbox_overlaps(gt, anchors): area1 = gt, area2 = anchor; union = area1 + area2 - overlap
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float overlap = w * h;
  float uni = area_a + area_b - overlap;
  uni = fmaxf(uni, 1e-6f);
  return overlap / uni;
}

struct AssignArgs {
  const float* anchors;     // [A,4]
  const uint8_t* valid;     // [B,A] or null (all valid)
  long long A;
  int B;
  const float* gts;         // [B,Gmax,4]
  const int* gt_count;      // [B]
  const long long* gt_labels;  // [B,Gmax]
  int Gmax;
  float pos_thr, neg_thr, min_pos_iou;
  int assign_all, num_classes;
  float means[4], stds[4];
  long long* assigned; long long* labels; float* label_w; float* bbox_t; float* bbox_w; int* num_pos;
  unsigned* gt_max_bits;    // ws [B,Gmax]
  unsigned* gt_argmax;      // ws [B,Gmax]
  int nlev;                 // 0: outputs are [B][A]; else level-major [L][B][A_l] (each level contiguous)
  long long lev_start[9];
};

__device__ __forceinline__ long long out_index(const AssignArgs& p, int b, long long a) {
  if (p.nlev == 0) return (long long)b * p.A + a;
  int l = 0;
  while (l + 1 < p.nlev && a >= p.lev_start[l + 1]) ++l;
  const long long s = p.lev_start[l], n = p.lev_start[l + 1] - s;
  return s * p.B + (long long)b * n + (a - s);
}

__global__ __launch_bounds__(256) void assign_pass1(const AssignArgs p) {
  __shared__ float sg[GMAX][5];
  __shared__ unsigned smax[GMAX];
  const int b = blockIdx.y;
  const int G = min(p.gt_count[b], p.Gmax);
  for (int i = threadIdx.x; i < G; i += 256) {
    const float* g = p.gts + ((long long)b * p.Gmax + i) * 4;
    sg[i][0] = g[0]; sg[i][1] = g[1]; sg[i][2] = g[2]; sg[i][3] = g[3];
    sg[i][4] = (g[2] - g[0]) * (g[3] - g[1]);
    smax[i] = 0u;
  }
  __syncthreads();
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a < p.A && G > 0 && (!p.valid || p.valid[(long long)b * p.A + a])) {
    const f32x4 an = *reinterpret_cast<const f32x4*>(p.anchors + a * 4);
    const float area = (an[2] - an[0]) * (an[3] - an[1]);
    for (int g = 0; g < G; ++g) {
      const float v = iou_ref(sg[g][0], sg[g][1], sg[g][2], sg[g][3], sg[g][4], an[0], an[1], an[2], an[3], area);
      atomicMax(&smax[g], __float_as_uint(v));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += 256)
    if (smax[i]) atomicMax(p.gt_max_bits + (long long)b * p.Gmax + i, smax[i]);
}

__global__ __launch_bounds__(256) void assign_pass2(const AssignArgs p) {
  __shared__ float sg[GMAX][5];
  __shared__ unsigned smx[GMAX], sarg[GMAX];
  const int b = blockIdx.y;
  const int G = min(p.gt_count[b], p.Gmax);
  for (int i = threadIdx.x; i < G; i += 256) {
    const float* g = p.gts + ((long long)b * p.Gmax + i) * 4;
    sg[i][0] = g[0]; sg[i][1] = g[1]; sg[i][2] = g[2]; sg[i][3] = g[3];
    sg[i][4] = (g[2] - g[0]) * (g[3] - g[1]);
    smx[i] = p.gt_max_bits[(long long)b * p.Gmax + i];
    sarg[i] = 0xffffffffu;
  }
  __syncthreads();
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a < p.A && G > 0 && (!p.valid || p.valid[(long long)b * p.A + a])) {
    const f32x4 an = *reinterpret_cast<const f32x4*>(p.anchors + a * 4);
    const float area = (an[2] - an[0]) * (an[3] - an[1]);
    for (int g = 0; g < G; ++g) {
      const float v = iou_ref(sg[g][0], sg[g][1], sg[g][2], sg[g][3], sg[g][4], an[0], an[1], an[2], an[3], area);
      if (__float_as_uint(v) == smx[g]) atomicMin(&sarg[g], (unsigned)a);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += 256)
    if (sarg[i] != 0xffffffffu) atomicMin(p.gt_argmax + (long long)b * p.Gmax + i, sarg[i]);
}

__global__ __launch_bounds__(256) void assign_pass3(const AssignArgs p) {
  __shared__ float sg[GMAX][5];
  __shared__ float smx[GMAX];
  __shared__ unsigned sarg[GMAX];
  __shared__ long long slab[GMAX];
  __shared__ int spos;
  const int b = blockIdx.y;
  const int G = min(p.gt_count[b], p.Gmax);
  if (threadIdx.x == 0) spos = 0;
  for (int i = threadIdx.x; i < G; i += 256) {
    const float* g = p.gts + ((long long)b * p.Gmax + i) * 4;
    sg[i][0] = g[0]; sg[i][1] = g[1]; sg[i][2] = g[2]; sg[i][3] = g[3];
    sg[i][4] = (g[2] - g[0]) * (g[3] - g[1]);
    smx[i] = __uint_as_float(p.gt_max_bits[(long long)b * p.Gmax + i]);
    sarg[i] = p.gt_argmax[(long long)b * p.Gmax + i];
    slab[i] = p.gt_labels[(long long)b * p.Gmax + i];
  }
  __syncthreads();
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  bool is_pos = false;
  if (a < p.A) {
    const long long o = out_index(p, b, a);
    const bool val = !p.valid || p.valid[(long long)b * p.A + a];
    long long asg = -1;
    const f32x4 an = *reinterpret_cast<const f32x4*>(p.anchors + a * 4);
    if (val) {
      if (G == 0) {
        asg = 0;
      } else {
        const float area = (an[2] - an[0]) * (an[3] - an[1]);
        float mx = -1.f;
        int arg = 0;
        for (int g = 0; g < G; ++g) {
          const float v = iou_ref(sg[g][0], sg[g][1], sg[g][2], sg[g][3], sg[g][4], an[0], an[1], an[2], an[3], area);
          if (v > mx) { mx = v; arg = g; }   // first maximal value, like torch.max(dim=0) on CPU
        }
        if (mx >= 0.f && mx < p.neg_thr) asg = 0;
        if (mx >= p.pos_thr) asg = arg + 1;
        for (int g = 0; g < G; ++g) {
          if (smx[g] >= p.min_pos_iou) {
            if (p.assign_all) {
              const float v = iou_ref(sg[g][0], sg[g][1], sg[g][2], sg[g][3], sg[g][4], an[0], an[1], an[2], an[3], area);
              if (v == smx[g]) asg = g + 1;
            } else if (sarg[g] == (unsigned)a) {
              asg = g + 1;
            }
          }
        }
      }
    }
    long long lab = p.num_classes;
    float lw = 0.f;
    f32x4 t = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
    if (asg > 0) {
      is_pos = true;
      const int g = (int)asg - 1;
      lab = slab[g];
      lw = 1.f;
      const float px = (an[0] + an[2]) * 0.5f, py = (an[1] + an[3]) * 0.5f, pw = an[2] - an[0], ph = an[3] - an[1];
      const float gx = (sg[g][0] + sg[g][2]) * 0.5f, gy = (sg[g][1] + sg[g][3]) * 0.5f, gw = sg[g][2] - sg[g][0], gh = sg[g][3] - sg[g][1];
      t[0] = ((gx - px) / pw - p.means[0]) / p.stds[0];
      t[1] = ((gy - py) / ph - p.means[1]) / p.stds[1];
      t[2] = (logf(gw / pw) - p.means[2]) / p.stds[2];
      t[3] = (logf(gh / ph) - p.means[3]) / p.stds[3];
      w = (f32x4){1.f, 1.f, 1.f, 1.f};
    } else if (asg == 0) {
      lw = 1.f;
    }
    p.assigned[o] = asg;
    p.labels[o] = lab;
    p.label_w[o] = lw;
    *reinterpret_cast<f32x4*>(p.bbox_t + o * 4) = t;
    *reinterpret_cast<f32x4*>(p.bbox_w + o * 4) = w;
  }
  const unsigned long long m = __ballot(is_pos);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&spos, __popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && spos) atomicAdd(p.num_pos + b, spos);
}

extern "C" size_t aod_assign_ws_bytes(int B, int Gmax) { return (size_t)B * Gmax * 8; }

extern "C" int aod_max_iou_assign(const float* anchors, const uint8_t* valid, int64_t A, int B, const float* gts, const int32_t* gt_count,
                                  const int64_t* gt_labels, int Gmax, float pos_thr, float neg_thr, float min_pos_iou,
                                  int gt_max_assign_all, int num_classes, const float* means4, const float* stds4, int64_t* assigned,
                                  int64_t* labels, float* label_w, float* bbox_t, float* bbox_w, int32_t* num_pos, void* ws,
                                  int nlev, const int64_t* level_start, aod_stream_t stream) {
  if (A == 0 || B == 0) return 0;
  AOD_CHECK_ARG(anchors && gts && gt_count && gt_labels && assigned && labels && label_w && bbox_t && bbox_w && num_pos && ws, "assign: null pointer");
  AOD_CHECK_ARG(Gmax >= 1 && Gmax <= GMAX, "assign: Gmax %d out of range (1..%d)", Gmax, GMAX);
  AssignArgs p;
  p.anchors = anchors; p.valid = valid; p.A = A; p.B = B; p.gts = gts; p.gt_count = gt_count; p.gt_labels = (const long long*)gt_labels;
  p.Gmax = Gmax; p.pos_thr = pos_thr; p.neg_thr = neg_thr; p.min_pos_iou = min_pos_iou; p.assign_all = gt_max_assign_all;
  p.num_classes = num_classes;
  for (int i = 0; i < 4; ++i) { p.means[i] = means4 ? means4[i] : 0.f; p.stds[i] = stds4 ? stds4[i] : 1.f; }
  p.assigned = (long long*)assigned; p.labels = (long long*)labels; p.label_w = label_w; p.bbox_t = bbox_t; p.bbox_w = bbox_w; p.num_pos = num_pos;
  AOD_CHECK_ARG(nlev >= 0 && nlev <= 8 && (nlev == 0 || level_start), "assign: bad level table");
  p.nlev = nlev;
  for (int i = 0; i <= nlev && nlev > 0; ++i) p.lev_start[i] = level_start[i];
  AOD_CHECK_ARG(nlev == 0 || (p.lev_start[0] == 0 && p.lev_start[nlev] == A), "assign: level table must cover [0, A)");
  p.gt_max_bits = (unsigned*)ws;
  p.gt_argmax = p.gt_max_bits + (size_t)B * Gmax;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(p.gt_max_bits, 0, (size_t)B * Gmax * 4, st);
  hipMemsetAsync(p.gt_argmax, 0xff, (size_t)B * Gmax * 4, st);
  hipMemsetAsync(num_pos, 0, (size_t)B * 4, st);
  dim3 grid((unsigned)((A + 255) / 256), B);
  hipLaunchKernelGGL(assign_pass1, grid, dim3(256), 0, st, p);
  if (!gt_max_assign_all) hipLaunchKernelGGL(assign_pass2, grid, dim3(256), 0, st, p);
  hipLaunchKernelGGL(assign_pass3, grid, dim3(256), 0, st, p);
  AOD_LAUNCH_CHECK();
  return 0;
}
