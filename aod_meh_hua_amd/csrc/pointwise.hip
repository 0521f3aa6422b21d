// Pointwise (1x1, stride 1, no padding) convolution = a plain GEMM  Y[M][N] = epi(X[M][K] . W[N][K]^T)  as a PERSISTENT STREAMING kernel.
//
// Why it exists: the bottleneck 1x1 convs of the backbone (32 of its 53 convs, and their dgrads) have K = 64 ... 2048 -- one to a few
// K-steps per output tile.  In the general implicit-GEMM kernel (conv.hip) a tile of such a layer spends 2 us in its prologue, 4-5 us until
// its first operand tile has landed, ~1.5 us in the K loop and 3 us in the epilogue (profiles/r01n_tile_phases.txt), all serial, two
// workgroups per CU: 1.2-2.3 TB/s on layers whose roofline is the HBM stream.  Here ONE workgroup per CU (8 waves) walks a contiguous
// range of tiles and treats their K-steps as one flat sequence:
//   * operands go global -> LDS by LDS-DMA into a 3-stage ring; the loads of step s+2 are issued at step s REGARDLESS of tile boundaries,
//     so the next tile's operands stream in under the current tile's MFMAs and stores (counted s_waitcnt vmcnt(N) + raw s_barrier: a
//     __syncthreads() would drain the ring at every step);
//   * the accumulators are computed TRANSPOSED (W fragment as the MFMA's A operand): a lane then holds 4 consecutive output channels of
//     one pixel, the epilogue runs in registers (scale / shift / residual / mask / ReLU) and stores 8 B per lane -- no LDS image, no
//     workgroup barrier, nothing that stops the ring;
//   * residual / mask tiles are fetched by loads the compiler does not see (inline asm, waited for by count): beside LDS-DMA hipcc
//     waits vmcnt(0) for every ordinary load, which would drain the ring once per tile;
//   * scale / shift vectors live in LDS for the whole launch, column sums are accumulated in LDS and flushed once per workgroup.
// vmcnt bookkeeping (MI355X_MICROARCH.md: loads, stores, LDS-DMA complete in issue order): every wait names how many YOUNGER vector
// memory operations may still be outstanding; stores and loads are issued unconditionally (rows past M use an out-of-range buffer
// offset), so the counts are exact.
#include "pointwise.h"

#include <stdlib.h>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector memory operations of this wave are outstanding (n is wave-uniform)
__device__ __forceinline__ void wait_vm_dyn(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break;
    case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break;
    case 13: wait_vm<13>(); break;
    case 14: wait_vm<14>(); break;
    case 15: wait_vm<15>(); break;
    case 16: wait_vm<16>(); break;
    case 17: wait_vm<17>(); break;
    case 18: wait_vm<18>(); break;
    case 19: wait_vm<19>(); break;
    case 20: wait_vm<20>(); break;
    case 21: wait_vm<21>(); break;
    case 22: wait_vm<22>(); break;
    case 23: wait_vm<23>(); break;
    case 24: wait_vm<24>(); break;
    case 25: wait_vm<25>(); break;
    case 26: wait_vm<26>(); break;
    case 27: wait_vm<27>(); break;
    case 28: wait_vm<28>(); break;
    case 29: wait_vm<29>(); break;
    case 30: wait_vm<30>(); break;
    case 31: wait_vm<31>(); break;
    case 32: wait_vm<32>(); break;
    case 33: wait_vm<33>(); break;
    case 34: wait_vm<34>(); break;
    case 35: wait_vm<35>(); break;
    case 36: wait_vm<36>(); break;
    case 37: wait_vm<37>(); break;
    case 38: wait_vm<38>(); break;
    case 39: wait_vm<39>(); break;
    case 40: wait_vm<40>(); break;
    case 41: wait_vm<41>(); break;
    case 42: wait_vm<42>(); break;
    case 43: wait_vm<43>(); break;
    case 44: wait_vm<44>(); break;
    case 45: wait_vm<45>(); break;
    case 46: wait_vm<46>(); break;
    case 47: wait_vm<47>(); break;
    default: wait_vm<48>(); break;      // (waiting for more than asked is always safe)
  }
}

// raw buffer descriptor in scalar registers for the inline-asm loads (same words as __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000))
__device__ __forceinline__ u32x4_t asm_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  u32x4_t d;
  d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  d[2] = __builtin_amdgcn_readfirstlane(bytes);
  d[3] = 0x00020000u;
  return d;
}

constexpr int PW_NMAX = 2048;          // columns whose scale / shift / column-sum vectors fit the LDS tables
constexpr int PW_NSTAGE = 3;

template <int BM, int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_gemm_kernel(const PwArgs p) {
  constexpr int BK = 64, ROWB = BK * 2, NSTAGE = PW_NSTAGE;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM / 64, B_IT = BN / 64, LPS = A_IT + B_IT;      // LDS-DMA instructions per wave and K-step
  constexpr int WM = BM / 4, WN = BN / 2, MI = WM / 16, NI = WN / 16;   // 8 waves as 4 (rows) x 2 (columns)
  constexpr int NP = MI * NI;                                            // 16 x 16 accumulator blocks per wave = 8-B pieces per lane
  static_assert(MI >= 1 && NI >= 1 && NP <= 8, "wave tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const s_scale = reinterpret_cast<float*>(smem + NSTAGE * STAGE);
  float* const s_shift = s_scale + PW_NMAX;
  float* const s_csum = s_shift + PW_NMAX;

  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = uw >> 1, wn = uw & 1;
  const int lr = lane & 15, lq = lane >> 4;

  for (int n = t; n < p.N; n += 512) {
    s_scale[n] = p.pre_scale ? p.pre_scale[n] : 1.f;
    s_shift[n] = p.pre_shift ? p.pre_shift[n] : 0.f;
    s_csum[n] = 0.f;
  }
  __syncthreads();          // (no LDS-DMA in flight yet: a plain barrier)

  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
  const int q0 = (int)((long long)blockIdx.x * ntiles / gridDim.x), q1 = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int nk = p.K / BK;
  const int total = (q1 - q0) * nk;
  const bool one = nk == 1;
  const bool has_res = p.res != nullptr, has_mask = p.mask != nullptr;
  const int R = NP * ((has_res ? 1 : 0) + (has_mask ? 1 : 0));     // hidden loads per tile
  constexpr int E = NP;                                             // stores per tile

  const unsigned x_bytes = (unsigned)((long long)p.M * p.K * 2), w_bytes = (unsigned)((long long)p.N * p.K * 2);
  const unsigned y_bytes = (unsigned)((long long)p.M * p.N * 2);
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
  const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)y_bytes, 0x00020000);
  const u32x4_t rsrc_res = asm_rsrc(p.res, has_res ? y_bytes : 0u), rsrc_mask = asm_rsrc(p.mask, has_mask ? y_bytes : 0u);
  constexpr unsigned OOB_BASE = 0xf0000000u;      // stays out of range under the per-step increments (operands < 3.5 GiB: host check)

  // ---- loader: the flat sequence of (tile, K-step) pairs of this workgroup
  // one LDS-DMA wave-instruction fills 8 tile rows x 8 16-B chunks linearly; the XOR swizzle that makes the ds_read_b128 fragment
  // reads conflict-free is applied on the source side: slot s of row r holds k-chunk s ^ ((r >> 1) & 7)
  const int prow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * uw + (lane >> 4)) & 7);
  int ld_q = q0, ld_kt = 0, istage = 0;
  unsigned aoff[A_IT], boff[B_IT];
  auto ld_tile = [&]() {
    const int tm = ld_q / tiles_n, tn = ld_q - tm * tiles_n;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = tm * BM + 64 * i + 8 * uw + prow;
      aoff[i] = m < p.M ? (unsigned)(((long long)m * p.K + kc * 8) * 2) : OOB_BASE;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int n = tn * BN + 64 * i + 8 * uw + prow;
      boff[i] = (unsigned)(((long long)n * p.K + kc * 8) * 2);
    }
  };
  auto issue = [&]() {
    char* sa = smem + istage * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const unsigned off = aoff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sa + (64 * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
      aoff[i] += ROWB;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const unsigned off = boff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sb + (64 * i + 8 * uw) * ROWB), 16, off, 0, 0, 0);
      boff[i] += ROWB;
    }
    istage = istage + 1 == NSTAGE ? 0 : istage + 1;
    if (++ld_kt == nk) {
      ld_kt = 0;
      if (++ld_q < q1) ld_tile();
    }
  };

  // ---- hidden residual / mask loads: 8 B per lane and accumulator block, the lane's 4 output channels of its pixel
  u32x2_t rres[NP], rmsk[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) { rres[k] = (u32x2_t){0u, 0u}; rmsk[k] = (u32x2_t){0u, 0u}; }
  auto piece_off = [&](int q, int i, int j) -> unsigned {      // byte offset of the lane's piece (i, j) of tile q in [M][N] bf16
    const int tm = q / tiles_n, tn = q - tm * tiles_n;
    const int m = tm * BM + wm * WM + i * 16 + lr, n = tn * BN + wn * WN + j * 16 + lq * 4;
    return m < p.M ? (unsigned)(((long long)m * p.N + n) * 2) : OOB_BASE;
  };
  auto issue_r = [&](int q) {
    if (has_res) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned off = piece_off(q, i, j);
          asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(rres[i * NI + j]) : "v"(off), "s"(rsrc_res) : "memory");
        }
    }
    if (has_mask) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned off = piece_off(q, i, j);
          asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(rmsk[i * NI + j]) : "v"(off), "s"(rsrc_mask) : "memory");
        }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: stages 0 and 1, then (single-K-step layers) the first tile's residual / mask
  if (total > 0) ld_tile();
  if (total > 0) issue();
  if (total > 1) issue();
  if (one && R && total > 0) issue_r(q0);

  int cq = q0, ckt = 0, cstage = 0;
  bool end_m1 = false, end_m2 = false;
  for (int s = 0; s < total; ++s) {
    const bool end_s = ckt == nk - 1;
    // operations younger than the loads of stage s that may stay in flight (file header): the next stage's loads, the stores of the
    // last two steps, the hidden loads issued since
    {
      const int nxt = s + 1 < total ? LPS : 0;
      const int n = one ? nxt + E * ((s >= 2 ? 1 : 0) + (s >= 1 ? 1 : 0)) + R * ((s >= 1 ? 1 : 0) + 1)
                        : nxt + E * ((end_m2 ? 1 : 0) + (end_m1 ? 1 : 0)) + (end_s ? R : 0);
      wait_vm_dyn(n);
    }
    __builtin_amdgcn_s_barrier();      // everybody's part of stage s has landed; everybody is done reading stage s - 1
    __builtin_amdgcn_sched_barrier(0);
    if (!one && R && ckt == nk - 2) issue_r(cq);       // operands of the epilogue one step ahead
    if (s + 2 < total) issue();                        // into the stage read at step s - 1
    __builtin_amdgcn_sched_barrier(0);

    const char* sa = smem + cstage * STAGE;
    const char* sb = sa + A_BYTES;
    cstage = cstage + 1 == NSTAGE ? 0 : cstage + 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 16 + lr;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + (((ks * 4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WN + j * 16 + lr;
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + (((ks * 4 + lq) ^ ((row >> 1) & 7)) << 4));
      }
      // transposed product: D[n][m] -- lane (lq, lr) ends up with channels 4 lq .. 4 lq + 3 of pixel lr
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }

    if (end_s) {
      if (R) {
        const int n = one ? (s + 2 < total ? LPS : 0) : (s + 1 < total ? LPS : 0) + (s + 2 < total ? LPS : 0);
        wait_vm_dyn(n);
        // (the hidden loads' destinations are opaque from here on: no consumer is scheduled above the wait)
#pragma unroll
        for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(rres[k]), "+v"(rmsk[k]));
      }
      const int tm = cq / tiles_n, tn = cq - tm * tiles_n;
      float cs[NI][4];
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = tm * BM + wm * WM + i * 16 + lr;
        const bool mok = m < p.M;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = tn * BN + wn * WN + j * 16 + lq * 4;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(s_scale + n), sh = *reinterpret_cast<const f32x4*>(s_shift + n);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * sc[r] + sh[r];
          if (has_res) {
            const u32x2_t rv = rres[i * NI + j];
            v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
            v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
          }
          if (has_mask) {
            const u32x2_t mv = rmsk[i * NI + j];
            v[0] = __uint_as_float(mv[0] << 16) > 0.f ? v[0] : 0.f; v[1] = __uint_as_float(mv[0] & 0xffff0000u) > 0.f ? v[1] : 0.f;
            v[2] = __uint_as_float(mv[1] << 16) > 0.f ? v[2] : 0.f; v[3] = __uint_as_float(mv[1] & 0xffff0000u) > 0.f ? v[3] : 0.f;
          }
          if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[j][r] += mok ? v[r] : 0.f;
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)v[r];
          const unsigned off = mok ? (unsigned)(((long long)m * p.N + n) * 2) : OOB_BASE;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ov), rsrc_y, (int)off, 0, 0);
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      if (p.colsum) {
        // sum over the 16 pixels of the lane group, then lane (lq, lr) keeps channel (lr >> 2) * 16 + 4 lq + (lr & 3): one LDS atomic per lane
        float pick = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = cs[j][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            pick = (lr == j * 4 + r) ? v : pick;
          }
        if ((lr >> 2) < NI) atomicAdd(s_csum + tn * BN + wn * WN + (lr >> 2) * 16 + lq * 4 + (lr & 3), pick);
      }
      ++cq;
      ckt = 0;
      if (one && R && cq < q1) issue_r(cq);
    } else ++ckt;
    end_m2 = end_m1;
    end_m1 = end_s;
  }

  if (p.colsum) {
    __syncthreads();
    for (int n = t; n < p.N; n += 512) {
      const float v = s_csum[n];
      if (v != 0.f) atomicAdd(p.colsum + n, v);
    }
  }
}

template <int BM, int BN>
int launch_pw(const PwArgs& a, hipStream_t st) {
  const size_t lds = (size_t)PW_NSTAGE * (BM + BN) * 128 + 3 * PW_NMAX * 4;
  static unsigned long long attr_done = 0;
  if (aod_first_on_device(&attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long ntiles = (long long)((a.M + BM - 1) / BM) * (a.N / BN);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL((pw_gemm_kernel<BM, BN>), dim3(grid), dim3(512), lds, st, a);
  return 0;
}

long long pw_tiles(const PwArgs& a, int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * (a.N / bn); }

}  // namespace

static int g_pw_mode = -1;
extern "C" int aod_set_pointwise_mode(int mode) {
  const int prev = g_pw_mode;
  g_pw_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  return prev;
}

bool aod_pw_wants(const PwArgs& a) {
  static const char* dbg = getenv("AOD_PW_STREAM");
  const int mode = g_pw_mode >= 0 ? g_pw_mode : ((dbg && (dbg[0] == '0' || dbg[0] == '1')) ? dbg[0] - '0' : -1);
  if (mode == 0) return false;
  if (a.colsum && aod_det_scratch(1)) return false;      // deterministic mode: column sums only through the general kernel's ordered partials
  if (a.K % 64 != 0 || a.N % 64 != 0 || a.N > PW_NMAX || a.M < 1) return false;
  if ((long long)a.M * a.K * 2 >= 0xe0000000ll || (long long)a.M * a.N * 2 >= 0xe0000000ll) return false;
  if (mode == 1) return true;
  // Measured against the general kernel on every 1x1 shape of the bench step (tools/dbg/pw_stream.py): both run the wide layers at the
  // HBM rate (3.7-5.2 TB/s) and the general kernel wins the deep, small-M ones (its 128 x 128 x 4-wave tiles at two workgroups per CU
  // hide more latency than one 8-wave workgroup); the streaming kernel wins where the output is narrow and long -- 64 columns over
  // >= 65536 rows: -9 % forward, -34 % in the dgrad form, whose column sums otherwise queue 4096 atomics per address.
  return a.N == 64 && a.M >= 65536;
}

int aod_pw_gemm(const PwArgs& a, hipStream_t st) {
  const int bn = a.N % 128 == 0 ? 128 : 64;
  // the larger row tile when its tiles still give every CU (nearly) the same number of them
  auto waste = [&](int bm) { const long long t = pw_tiles(a, bm, bn); return (double)(((t + 255) / 256) * 256) / (double)t; };
  const bool big = pw_tiles(a, 128, bn) >= 256 && waste(128) <= waste(64) * 1.05;
  if (bn == 128) { if (big) launch_pw<128, 128>(a, st); else launch_pw<64, 128>(a, st); }
  else { if (big) launch_pw<128, 64>(a, st); else launch_pw<64, 64>(a, st); }
  AOD_LAUNCH_CHECK();
  return 0;
}
