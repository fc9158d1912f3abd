// Per-CU global -> LDS DMA (global_load_lds_dwordx4) ingest-rate microbenchmark for gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dma_bench tools/dma_bench.cpp
// One workgroup per CU streams 16-B-per-lane copies into an LDS ring with a bounded number of copies in flight per
// wave (s_waitcnt vmcnt(DEPTH)).  Reported: GB/s per CU and aggregate, for
//   footprint  = bytes each block cycles through (its own window; "shared" = every block reads the same window)
//   waves      = 4 / 8 / 16 per CU,   depth = copies in flight per wave
// Purpose: the feed rate that bounds the GEMM main loop (72 KiB per 256x320x64 K-tile per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_wave_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_addr), "v"(voff),
               "s"(sbase)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// rows of 128 B; a wave copy covers 8 rows x 128 B (like the GEMM staging); `row_stride` bytes between rows.
// miss_every = n > 0: every n-th copy of a wave reads from the far (HBM) region instead of the block's window.
// reader_waves = r: the LAST r waves of the block do not copy; they hammer the LDS with ds_read_b128 instead.
template <int DEPTH, bool TO_LDS>
__global__ void dma_kernel(const char* src, size_t window, size_t block_stride, int row_stride, int iters, float* sink,
                           int miss_every, const char* far, size_t far_block, int reader_waves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = (blockDim.x >> 6) - reader_waves;
  if (wave >= nwaves) {
    // LDS reader: 16 conflict-free ds_read_b128 per iteration
    float a = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v;
        const uint32_t ad = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + ((k * 1024 + lane * 16) & 16383);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        a += 0.f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (sink != nullptr && a == 123.f) sink[1] = a;
    return;
  }
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 4 * 1024;
  const char* base = src + (size_t)blockIdx.x * block_stride;
  const char* fbase = far + (size_t)blockIdx.x * far_block;
  // lane -> (row, chunk) within the wave's 8-row group
  const uint32_t lane_off = (uint32_t)(lane >> 3) * row_stride + (lane & 7) * 16;
  const size_t group_bytes = (size_t)8 * row_stride;          // one wave copy
  const size_t step = group_bytes * nwaves;                     // all waves of the block
  size_t pos = (size_t)wave * group_bytes, fpos = (size_t)wave * 1024;
  int slot = 0, mc = 0;
  for (int it = 0; it < iters; ++it) {
    const char* s = base + pos;
    uint32_t lo = lane_off;
    if (miss_every > 0 && ++mc == miss_every) {
      mc = 0;
      s = fbase + fpos;
      lo = (uint32_t)(lane >> 3) * 128 + (lane & 7) * 16;
      fpos += (size_t)nwaves * 1024;
      if (fpos + 1024 > far_block) fpos = (size_t)wave * 1024;
    }
    if (TO_LDS) {
      glds16_s(s, lo, lds0 + slot * 1024);
      wait_vm<DEPTH>();
    } else {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lo), "s"(s) : "memory");
      wait_vm<DEPTH>();
    }
    pos += step;
    if (pos + group_bytes > window) pos = (size_t)wave * group_bytes;
    slot = (slot + 1) & 3;
  }
  wait_vm<0>();
}

struct Cfg { size_t window, stride; int row_stride, threads, miss_every, readers; };

template <int DEPTH, bool TO_LDS>
static double run(const char* src, const char* far, size_t far_block, const Cfg& c, int blocks, int iters, hipStream_t st) {
  size_t lds = 64 * 1024;
  auto kern = dma_kernel<DEPTH, TO_LDS>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(c.threads), lds, st, src, c.window, c.stride, c.row_stride, iters / 4,
                     (float*)nullptr, c.miss_every, far, far_block, c.readers);
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(c.threads), lds, st, src, c.window, c.stride, c.row_stride, iters,
                     (float*)nullptr, c.miss_every, far, far_block, c.readers);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)blocks * (c.threads / 64 - c.readers) * iters * 1024.0;
  return bytes / (ms * 1e-3) * 1e-9;   // GB/s aggregate
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  // one allocation: [0, 96 MiB) = per-block windows (L2-resident cases), [96 MiB, 1.1 GiB) = the "far" region whose
  // lines are (almost) never in L2 when touched
  const size_t near_bytes = (size_t)96 << 20, far_bytes = (size_t)1 << 30;
  char* src;
  CK(hipMalloc(&src, near_bytes + far_bytes));
  CK(hipMemset(src, 1, near_bytes + far_bytes));
  CK(hipDeviceSynchronize());
  char* far = src + near_bytes;
  const size_t far_block = far_bytes / 256;
  struct Case { const char* name; Cfg c; };
  const size_t K64 = 64 << 10;
  Case cases[] = {
      {"L2 64K/blk contiguous rows, 8w", {K64, K64, 128, 512, 0, 0}},
      {"L2 64K/blk row stride 640 B, 8w", {K64 * 5, K64 * 5, 640, 512, 0, 0}},
      {"L2 shared 1.8M window, row stride 5760 B, 8w", {(size_t)320 * 5760, 0, 5760, 512, 0, 0}},
      {"L2 64K/blk contiguous, 1 of 9 copies from HBM, 8w", {K64, K64, 128, 512, 9, 0}},
      {"L2 64K/blk contiguous, 1 of 3 copies from HBM, 8w", {K64, K64, 128, 512, 3, 0}},
      {"L2 64K/blk contiguous, 4 copy + 4 LDS-reader waves", {K64, K64, 128, 512, 0, 4}},
      {"L2 64K/blk contiguous, 8 copy + 8 LDS-reader waves", {K64, K64, 128, 1024, 0, 8}},
      {"L2 64K/blk stride 640, 1 of 9 HBM, 4 copy + 4 readers", {K64 * 5, K64 * 5, 640, 512, 9, 4}},
  };
  printf("%-58s %6s %6s %12s %12s\n", "case", "depth", "mode", "GB/s per CU", "TB/s total");
  const int iters = 4096;
  for (auto& cs : cases) {
    printf("# %s\n", cs.name);
    double r4 = run<4, true>(src, far, far_block, cs.c, cus, iters, st);
    double r16 = run<16, true>(src, far, far_block, cs.c, cus, iters, st);
    double v16 = run<16, false>(src, far, far_block, cs.c, cus, iters, st);
    printf("%-58s %6d %6s %12.1f %12.2f\n", cs.name, 4, "lds", r4 / cus, r4 * 1e-3);
    printf("%-58s %6d %6s %12.1f %12.2f\n", cs.name, 16, "lds", r16 / cus, r16 * 1e-3);
    printf("%-58s %6d %6s %12.1f %12.2f\n", cs.name, 16, "vgpr", v16 / cus, v16 * 1e-3);
  }
  return 0;
}
