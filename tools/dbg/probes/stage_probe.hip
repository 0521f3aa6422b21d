// How fast does ONE CU get operand tiles from L2 into LDS?  Three paths, same addresses, same bytes:
//   0  LDS-DMA            buffer_load_dwordx4 ... lds   (what conv.hip / the fused kernels stage with)
//   1  VGPR + ds_write    buffer_load_dwordx4 -> registers -> ds_write_b128
//   2  VGPR only          buffer_load_dwordx4 -> registers (xor-folded, no LDS write): the L2 -> CU rate itself
// Each workgroup (512 threads) streams `iters` tiles of 64 KB out of a window of `win` bytes (L2-resident for win <= a few MB; HBM-sized
// otherwise), `depth` tiles in flight.  Prints bytes per shader clock per CU (s_memtime) and GB/s per CU.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/stage_probe tools/dbg/probes/stage_probe.hip && /tmp/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, unsigned win, int iters, unsigned long long* out, unsigned* sink, int bcast, unsigned rstride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // DEPTH x 64 KB
  const int t = threadIdx.x, lane = t & 63;
  const int uw = __builtin_amdgcn_readfirstlane(t >> 6);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)win, 0x00020000);
  // tile k of this workgroup starts at ((blockIdx * 7 + k) * 64 KB) mod win; a wave moves 8 x 1 KB pieces of it
  // bcast: every workgroup walks the SAME tiles (a filter: all CUs read the same lines at about the same time);
  // rstride: a 1 KB piece is 8 rows of 128 B `rstride` bytes apart (a [N][K] filter matrix read 64 columns at a time) instead of 1 KB contiguous
  unsigned base = bcast ? 0u : (unsigned)(((unsigned long long)blockIdx.x * 7u * 65536u) % win);
  u32x4 fold = {0, 0, 0, 0};
  auto issue = [&](int k, int slot, u32x4 (&r)[8]) {
    const unsigned off0 = (unsigned)(((unsigned long long)base + (unsigned long long)k * 65536u) % win);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned off = rstride ? (unsigned)(((unsigned long long)off0 / 65536u * 128u + (unsigned long long)((uw * 8 + i) * 8 + (lane >> 3)) * rstride + (lane & 7) * 16) % win)
                                   : off0 + (unsigned)((uw * 8 + i) * 1024 + lane * 16);
      if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + slot * 65536 + (uw * 8 + i) * 1024), 16, off, 0, 0, 0);
      else r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
    }
  };
  u32x4 regs[DEPTH][8];
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d, regs[d]);
  for (int k = 0; k < iters; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // tile k + d has landed when only the DEPTH - 1 younger tiles (8 instructions each) are still out
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + d * 65536 + (uw * 8 + i) * 1024 + lane * 16) = regs[d][i];
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fold ^= regs[d][i];
      }
      __builtin_amdgcn_s_barrier();                       // (a consumer would read the tile here)
      issue(k + DEPTH + d, d, regs[d]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (MODE != 2) fold[0] ^= *reinterpret_cast<unsigned*>(smem + (t * 16) % 65536);
  if ((fold[0] ^ fold[1] ^ fold[2] ^ fold[3]) == 0x12345678u) sink[0] = fold[0];
  if (t == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, unsigned win, int wgs, int iters, unsigned long long* out, unsigned* sink, int bcast = 0, unsigned rstride = 0) {
  const size_t lds = (size_t)DEPTH * 65536;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(wgs), dim3(512), lds, 0, src, win, iters, out, sink, bcast, rstride);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * wgs);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, tick = 0;
  for (int i = 0; i < wgs; ++i) { cyc += (double)h[2 * i]; tick += (double)h[2 * i + 1]; }
  cyc /= wgs; tick /= wgs;
  const double bytes = (double)(iters + DEPTH) * 65536.0;
  printf("  %-18s %s%s depth %d  window %6.1f MB  %4d wgs: %6.1f B/clk/CU  %6.1f GB/s/CU  (clock %.2f GHz)  aggregate %5.2f TB/s\n", name, bcast ? "[same tiles in every CU] " : "", rstride ? "[rows 9216 B apart] " : "", DEPTH, win / 1048576.0, wgs,
         bytes / cyc, bytes / (tick * 10.0), cyc / (tick * 10.0), bytes / (tick * 10.0) * wgs / 1000.0);
}

int main() {
  const size_t cap = 1ull << 30;
  char* src; unsigned long long* out; unsigned* sink;
  hipMalloc(&src, cap); hipMemset(src, 1, cap);
  hipMalloc(&out, 8 * 2 * 1024); hipMalloc(&sink, 64);
  const int iters = 600;
  for (unsigned win : {2u << 20, 16u << 20, 1u << 30}) {
    for (int wgs : {256, 64}) {
      printf("window %.0f MB, %d workgroups (one per CU):\n", win / 1048576.0, wgs);
      run<0, 2>("LDS-DMA", src, win, wgs, iters, out, sink);
      run<1, 2>("VGPR + ds_write", src, win, wgs, iters, out, sink);
      run<2, 2>("VGPR only", src, win, wgs, iters, out, sink);
      if (wgs == 256) {
        run<0, 1>("LDS-DMA", src, win, wgs, iters, out, sink);
        run<2, 1>("VGPR only", src, win, wgs, iters, out, sink);
      }
    }
  }
  printf("filter-like access (LDS-DMA, two tiles in flight), 256 workgroups:\n");
  run<0, 2>("LDS-DMA", src, 4u << 20, 256, iters, out, sink, 1, 0);
  run<0, 2>("LDS-DMA", src, 4u << 20, 256, iters, out, sink, 0, 9216);
  run<0, 2>("LDS-DMA", src, 4u << 20, 256, iters, out, sink, 1, 9216);
  run<0, 2>("LDS-DMA", src, 4u << 20, 256, iters, out, sink, 1, 9216 + 128);
  run<0, 2>("LDS-DMA", src, 4u << 20, 256, iters, out, sink, 1, 1024);
  return 0;
}
