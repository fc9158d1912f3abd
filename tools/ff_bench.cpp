// Stand-alone benchmark + check of the fused GEGLU feed-forward prototype (vx_ff_fused) against the product's two
// launches (vx_gemm GEGLU with the folded LayerNorm, then vx_gemm STORE with the residual) on the 64x64-level shape.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ff_bench tools/ff_bench.cpp -ldl
//   tools/ff_bench v-express_amd/libvexpress_hip.so [reps=20] [rows=131072]
// Prints us per launch (pair / fused), TFLOP/s of the 2 x rows x (2560 x 320 + 320 x 1280) FLOP, and the difference of the
// two results over ALL elements (the arithmetic is the same chain in the same order: expected to agree to the last bf16 bit
// almost everywhere).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vexpress_hip.h"

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

typedef uint16_t bf16_t;
__host__ __device__ static inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
__host__ __device__ static inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__host__ __device__ static inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void fill_bf16(bf16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = f2bf(((int)(hash32((uint32_t)i * 2654435761u + seed) & 0xffff) - 32768) * (scale / 32768.0f));
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale, float offset) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = offset + ((int)(hash32((uint32_t)i + seed) & 0xffff) - 32768) * (scale / 32768.0f);
}
// ln_stats[m] = (mean, rstd): mean in [-0.25, 0.25], rstd in [0.75, 1.25]
__global__ void fill_stats(float* p, size_t rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) {
    p[2 * i] = ((int)(hash32((uint32_t)i + 77u) & 0xffff) - 32768) * (0.25f / 32768.0f);
    p[2 * i + 1] = 1.0f + ((int)(hash32((uint32_t)i + 99u) & 0xffff) - 32768) * (0.25f / 32768.0f);
  }
}
__global__ void diff_kernel(const bf16_t* a, const bf16_t* b, size_t n, float* out /* [maxabsdiff, maxabsref, ndiff] */) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float md = 0.f, mr = 0.f, nd = 0.f;
  for (; i < n; i += stride) {
    const float x = bf2f(a[i]), y = bf2f(b[i]);
    md = fmaxf(md, fabsf(x - y));
    mr = fmaxf(mr, fabsf(y));
    nd += a[i] != b[i] ? 1.f : 0.f;
  }
  atomicMax((int*)&out[0], __float_as_int(md));
  atomicMax((int*)&out[1], __float_as_int(mr));
  atomicAdd(&out[2], nd);
}

typedef int (*gemm_fn)(const vx_gemm_params*, void*);
typedef int (*ff_fn)(const vx_ff_params*, void*);
typedef int (*pack_fn)(const void*, const void*, void*, void*, int, int, void*);
typedef const char* (*err_fn)(void);

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <libvexpress_hip.so> [reps=20] [rows=131072]\n", argv[0]);
    return 1;
  }
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  gemm_fn gemm = (gemm_fn)dlsym(lib, "vx_gemm");
  ff_fn ff = (ff_fn)dlsym(lib, "vx_ff_fused");
  pack_fn pack = (pack_fn)dlsym(lib, "vx_ff_pack_weights");
  err_fn lasterr = (err_fn)dlsym(lib, "vx_last_error_string");
  if (!gemm || !ff || !pack) {
    fprintf(stderr, "library lacks vx_gemm / vx_ff_fused / vx_ff_pack_weights\n");
    return 1;
  }
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int M = argc > 3 ? atoi(argv[3]) : 131072, C = 320, H = 1280;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  bf16_t *x, *w1, *w2, *w1t, *w2t, *g, *yref, *y;
  float *b1, *cs, *stats, *b2, *d;
  CK(hipMalloc(&x, (size_t)M * C * 2)); CK(hipMalloc(&w1, (size_t)2 * H * C * 2)); CK(hipMalloc(&w2, (size_t)C * H * 2));
  CK(hipMalloc(&w1t, (size_t)2 * H * C * 2)); CK(hipMalloc(&w2t, (size_t)C * H * 2));
  CK(hipMalloc(&g, (size_t)M * H * 2)); CK(hipMalloc(&yref, (size_t)M * C * 2)); CK(hipMalloc(&y, (size_t)M * C * 2));
  CK(hipMalloc(&b1, 2 * H * 4)); CK(hipMalloc(&cs, 2 * H * 4)); CK(hipMalloc(&stats, (size_t)M * 2 * 4)); CK(hipMalloc(&b2, C * 4));
  CK(hipMalloc(&d, 3 * 4));
  fill_bf16<<<2048, 256, 0, st>>>(x, (size_t)M * C, 1u, 1.0f);
  fill_bf16<<<2048, 256, 0, st>>>(w1, (size_t)2 * H * C, 2u, 1.0f / sqrtf((float)C));
  fill_bf16<<<2048, 256, 0, st>>>(w2, (size_t)C * H, 3u, 1.0f / sqrtf((float)H));
  fill_f32<<<(2 * H + 255) / 256, 256, 0, st>>>(b1, 2 * H, 4u, 0.5f, 0.f);
  fill_f32<<<(2 * H + 255) / 256, 256, 0, st>>>(cs, 2 * H, 5u, 0.5f, 0.f);
  fill_f32<<<(C + 255) / 256, 256, 0, st>>>(b2, C, 6u, 0.5f, 0.f);
  fill_stats<<<(M + 255) / 256, 256, 0, st>>>(stats, M);
  CK(hipMemsetAsync(y, 0, (size_t)M * C * 2, st));
  if (pack(w1, w2, w1t, w2t, C, H, st)) { fprintf(stderr, "pack: %s\n", lasterr()); return 1; }

  vx_gemm_params p1, p2;
  memset(&p1, 0, sizeof(p1));
  p1.a = x; p1.c1 = C; p1.lda1 = C; p1.nb = 1; p1.h_in = M; p1.w_in = 1; p1.kh = p1.kw = 1; p1.stride = 1;
  p1.h_out = M; p1.w_out = 1; p1.w = w1; p1.n = 2 * H; p1.k = C; p1.m = M; p1.alpha = 1.f; p1.bias = b1;
  p1.epi = VX_EPI_GEGLU; p1.out = g; p1.ldc = H; p1.ln_stats = stats; p1.ln_colsum = cs; p1.ring_hint = 1;
  memset(&p2, 0, sizeof(p2));
  p2.a = g; p2.c1 = H; p2.lda1 = H; p2.nb = 1; p2.h_in = M; p2.w_in = 1; p2.kh = p2.kw = 1; p2.stride = 1;
  p2.h_out = M; p2.w_out = 1; p2.w = w2; p2.n = C; p2.k = H; p2.m = M; p2.alpha = 1.f; p2.bias = b2;
  p2.epi = VX_EPI_STORE; p2.out = yref; p2.ldc = C; p2.residual = x; p2.ldr = C; p2.ring_hint = 1;
  vx_ff_params pf;
  memset(&pf, 0, sizeof(pf));
  pf.x = x; pf.ldx = C; pf.m = M; pf.c = C; pf.hidden = H; pf.w1t = w1t; pf.w2t = w2t; pf.bias1 = b1; pf.ln_colsum = cs;
  pf.ln_stats = stats; pf.bias2 = b2; pf.residual = x; pf.ldr = C; pf.out = y; pf.ldo = C;

  if (gemm(&p1, st) || gemm(&p2, st)) { fprintf(stderr, "vx_gemm: %s\n", lasterr()); return 1; }
  if (ff(&pf, st)) { fprintf(stderr, "vx_ff_fused: %s\n", lasterr()); return 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemsetAsync(d, 0, 12, st));
  diff_kernel<<<1024, 256, 0, st>>>(y, yref, (size_t)M * C, d);
  float hd[3];
  CK(hipMemcpyAsync(hd, d, 12, hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  printf("fused vs two launches over %zu elements: max|diff| %.4g, max|ref| %.4g, differing elements %.0f (%.3g %%)  %s\n",
         (size_t)M * C, hd[0], hd[1], hd[2], 100.0 * hd[2] / ((double)M * C),
         hd[0] <= hd[1] / 128 && hd[1] > 0 ? "ok" : "MISMATCH");

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double fl = 2.0 * M * ((double)2 * H * C + (double)C * H);
  for (int round = 0; round < 3; ++round) {
    float ms_pair, ms_ff, ms_g;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) { gemm(&p1, st); gemm(&p2, st); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_pair, e0, e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) gemm(&p1, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_g, e0, e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) ff(&pf, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ff, e0, e1));
    printf("round %d: two launches %.1f us (GEGLU %.1f + FF-out %.1f) = %.1f TF/s | fused %.1f us = %.1f TF/s\n", round,
           1e3 * ms_pair / reps, 1e3 * ms_g / reps, 1e3 * (ms_pair - ms_g) / reps, fl / (1e3 * ms_pair / reps) * 1e-6,
           1e3 * ms_ff / reps, fl / (1e3 * ms_ff / reps) * 1e-6);
  }
  return 0;
}
