// Deterministic mode (aod_set_deterministic): the bias / BN-shift column sums of the training path -- its only order-dependent reduction -- are
// formed from per-workgroup PARTIALS added in a fixed order instead of fp32 atomics (arrival order).  The producing kernels (conv.hip
// epilogue and split-K finalize, x3_ops.hip / elementwise.hip activation-backward and pad-cast passes) store their partial column sums as
// rows of a per-device scratch; colsum_finalize_kernel then adds the rows top to bottom into the destination vector.  Costs one small launch per
// column-sum vector (~4 us in a replayed graph, ~1 % of a training step); off by default.  Same role as torch.use_deterministic_algorithms
// for the reference's `--deterministic` switch (tools/train_RetinaNet.py:56-68 -> mmdet/apis/train.py set_random_seed).
#include "common.h"

namespace {
constexpr size_t SCRATCH_FLOATS = 8u << 20;        // 32 MB: the largest user is M / 64 tiles x N columns (65 536 / 64 x 512 = 2 MB)
int g_det = 0;
float* g_scratch[64] = {nullptr};

// block = 16 columns x 16 lanes; lane l adds the partial rows l, l + 16, ... (four independent chains in flight: the rows are tiny, the cost
// is load latency), the 16 lane sums are added in lane order -- a fixed tree, whatever finished first
__device__ __forceinline__ float ordered_colsum(const float* __restrict__ ws, int nparts, int pitch, int n, bool ok, float (*sm)[17]) {
  const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ok) {
    int p = lane;
    for (; p + 48 < nparts; p += 64) {
      a0 += ws[(long long)p * pitch + n]; a1 += ws[(long long)(p + 16) * pitch + n];
      a2 += ws[(long long)(p + 32) * pitch + n]; a3 += ws[(long long)(p + 48) * pitch + n];
    }
    for (; p < nparts; p += 16) a0 += ws[(long long)p * pitch + n];
  }
  sm[lane][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float s = 0.f;
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < 16; ++l) s += sm[l][col];
  }
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ ws, int nparts, int pitch, int N, float* __restrict__ dst,
                                                              float* __restrict__ dst2, long long ws2_off) {
  __shared__ float sm[16][17];
  const int n = blockIdx.x * 16 + (threadIdx.x & 15);
  const bool ok = n < N;
  const float s = ordered_colsum(ws, nparts, pitch, n, ok, sm);
  if (ok && threadIdx.x < 16) dst[n] += s;
  if (dst2) {
    const float s2 = ordered_colsum(ws + ws2_off, nparts, pitch, n, ok, sm);
    if (ok && threadIdx.x < 16) dst2[n] += s2;
  }
}
}  // namespace

extern "C" int aod_set_deterministic(int on) {
  const int prev = g_det;
  if (on) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return aod_set_err(-3, "aod_set_deterministic: no current device");
    if (!g_scratch[d]) {
      if (hipMalloc((void**)&g_scratch[d], SCRATCH_FLOATS * sizeof(float)) != hipSuccess) return aod_set_err(-3, "aod_set_deterministic: scratch allocation failed");
    }
  }
  g_det = on ? 1 : 0;
  return prev;
}

extern "C" int aod_get_deterministic(void) { return g_det; }

// scratch of the current device for `floats` partial values, or nullptr when the mode is off (callers then take their atomic path).
// With the mode ON the scratch is allocated lazily per device (the switch is process-wide, a second device may never have seen the call) and
// a request that does not fit is REPORTED once: the caller's atomic fallback is still correct, but the run is no longer bit-repeatable
// (ADVICE r4).  One scratch per device, shared by every launch: deterministic training runs on ONE stream per device (include/aod_hip.h).
float* aod_det_scratch(size_t floats) {
  if (!g_det) return nullptr;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return nullptr;
  if (!g_scratch[d]) {
    // lazily, for a device that never saw aod_set_deterministic.  The first request may come from inside a stream capture, where a plain
    // hipMalloc is an illegal call that invalidates the graph being built: the allocation runs under the RELAXED capture mode of
    // this thread (what torch's caching allocator does around its own hipMalloc), which permits it and leaves the capture intact
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (hipMalloc((void**)&g_scratch[d], SCRATCH_FLOATS * sizeof(float)) != hipSuccess) g_scratch[d] = nullptr;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
  }
  if (!g_scratch[d] || floats > SCRATCH_FLOATS) {
    static unsigned long long warned = 0;             // one bit per device
    const unsigned long long bit = 1ull << d;
    if (!(__atomic_fetch_or(&warned, bit, __ATOMIC_RELAXED) & bit)) {
      fprintf(stderr, "[libaodhip] deterministic mode: %zu partial sums do not fit the %zu-float scratch of device %d (or it could not be allocated -- "
                      "call aod_set_deterministic(1) on every device BEFORE capturing graphs): this launch falls back to fp32 atomics and the run is NOT "
                      "bit-repeatable\n", floats, (size_t)SCRATCH_FLOATS, d);
    }
    return nullptr;
  }
  return g_scratch[d];
}

// dst[n] += sum_p ws[p][n]  (dst2 likewise from the second block of partials at ws + ws2_off), rows in order
int aod_colsum_finalize(const float* ws, int nparts, int pitch, int N, float* dst, float* dst2, long long ws2_off, hipStream_t st) {
  if (nparts <= 0 || N <= 0) return 0;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((N + 15) / 16), dim3(256), 0, st, ws, nparts, pitch, N, dst, dst2, ws2_off);
  AOD_LAUNCH_CHECK();
  return 0;
}
