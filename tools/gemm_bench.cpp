// Stand-alone per-shape benchmark + spot-check of vx_gemm (no torch: starts in seconds on the GPU box).
//   hipcc --offload-arch=gfx950 -O2 -o tools/gemm_bench tools/gemm_bench.cpp -ldl
//   tools/gemm_bench v-express_amd/libvexpress_hip.so [reps] [filter]
// For each GEMM / implicit-conv shape of the UNet3D CFG forward at 512^2 f=16 (and the VAE decode) it
//   * fills A / W with a hash-based pattern on the device,
//   * checks 4096 sampled outputs against a naive fp32 device reference (conv gather restated independently),
//   * times `reps` launches with HIP events and prints us / TFLOP/s.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

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
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
    p[i] = f2bf(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = ((int)(hash32((uint32_t)i + seed) & 0xffff) - 32768) * (scale / 32768.0f);
}

struct RefArgs {
  const bf16_t *a, *a2, *w;
  const float* bias;
  const bf16_t* residual;
  int c1, c2, lda1, lda2, nb, h_in, w_in, kh, kw, stride, pad, up, h_out, w_out, n, k, m, ldr;
  int geglu;   // 1: sample column o of the GEGLU output = value row 16 (o / 8) + o % 8 times erf-GELU of gate row + 8
};
// one thread per sampled (m, n): plain fp32 dot product over the gathered row
__global__ void ref_samples(RefArgs r, const int* sm, const int* sn, float* out, int ns) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  int m = sm[s], n = sn[s];
  if (r.geglu) {
    // plain linear (kh = kw = 1): two dot products
    const int vr = 16 * (n / 8) + n % 8, gr = vr + 8;
    float av = 0.f, ag = 0.f;
    for (int c = 0; c < r.c1; ++c) {
      const float x = bf2f(r.a[(size_t)m * r.lda1 + c]);
      av += x * bf2f(r.w[(size_t)vr * r.k + c]);
      ag += x * bf2f(r.w[(size_t)gr * r.k + c]);
    }
    if (r.bias) { av += r.bias[vr]; ag += r.bias[gr]; }
    out[s] = av * 0.5f * ag * (1.0f + erff(ag * 0.70710678f));
    return;
  }
  int hw = r.h_out * r.w_out;
  int fr = m / hw, rem = m % hw, oy = rem / r.w_out, ox = rem % r.w_out;
  int cin = r.c1 + r.c2;
  int he = r.h_in << r.up, we = r.w_in << r.up;
  float acc = 0.f;
  for (int ky = 0; ky < r.kh; ++ky)
    for (int kx = 0; kx < r.kw; ++kx) {
      int iy = oy * r.stride - r.pad + ky, ix = ox * r.stride - r.pad + kx;
      if (iy < 0 || iy >= he || ix < 0 || ix >= we) continue;
      size_t pix = (size_t)fr * r.h_in * r.w_in + (size_t)(iy >> r.up) * r.w_in + (ix >> r.up);
      const bf16_t* wrow = r.w + (size_t)n * r.k + (size_t)(ky * r.kw + kx) * cin;
      for (int c = 0; c < r.c1; ++c) acc += bf2f(r.a[pix * r.lda1 + c]) * bf2f(wrow[c]);
      for (int c = 0; c < r.c2; ++c) acc += bf2f(r.a2[pix * r.lda2 + c]) * bf2f(wrow[r.c1 + c]);
    }
  if (r.bias) acc += r.bias[n];
  if (r.residual) acc += bf2f(r.residual[(size_t)m * r.ldr + n]);
  out[s] = acc;
}
__global__ void gather_out(const bf16_t* out, int ldc, const int* sm, const int* sn, float* o, int ns) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < ns) o[s] = bf2f(out[(size_t)sm[s] * ldc + sn[s]]);
}

struct Shape {
  const char* name;
  int nb, h, w, c1, c2, ks, stride, up, n, epi;   // epi: 0 store(+bias), 1 geglu, 2 split(qkv), 3 store + residual
  int per_fwd;                                    // launches of this shape per CFG forward (approximate; for weights)
  int prepad;                                     // 1: h, w already include a zero border, conv runs with pad 0
};

static const Shape SHAPES[] = {
    // ---- UNet3D, level 0: 32 frames x 64x64, C=320
    {"L0 conv3x3 320>320", 32, 64, 64, 320, 0, 3, 1, 0, 320, 0, 7},
    {"L0 conv3x3 640cat>320", 32, 64, 64, 320, 320, 3, 1, 0, 320, 0, 2},
    {"L0 conv3x3 960cat>320", 32, 64, 64, 640, 320, 3, 1, 0, 320, 0, 1},
    {"L0 lin 320>320 +res", 32, 64, 64, 320, 0, 1, 1, 0, 320, 3, 40},
    {"L0 qkv 320>960 split", 32, 64, 64, 320, 0, 1, 1, 0, 960, 2, 15},
    {"L0 geglu 320>2560", 32, 64, 64, 320, 0, 1, 1, 0, 2560, 1, 10},
    {"L0 ffout 1280>320 +res", 32, 64, 64, 1280, 0, 1, 1, 0, 320, 3, 10},
    {"L0 down s2 320>320", 32, 64, 64, 320, 0, 3, 2, 0, 320, 0, 1},
    {"L0 conv3x3 320>320 prepad", 32, 66, 66, 320, 0, 3, 1, 0, 320, 0, 0, 1},
    {"L0 conv3x3 640cat>320 prepad", 32, 66, 66, 320, 320, 3, 1, 0, 320, 0, 0, 1},
    {"L1 conv3x3 640>640 prepad", 32, 34, 34, 640, 0, 3, 1, 0, 640, 0, 0, 1},
    {"L2 conv3x3 1280>1280 prepad", 32, 18, 18, 1280, 0, 3, 1, 0, 1280, 0, 0, 1},
    {"L3 conv3x3 1280>1280 prepad", 32, 10, 10, 1280, 0, 3, 1, 0, 1280, 0, 0, 1},
    {"VAE 512^2 128>128 prepad", 4, 514, 514, 128, 0, 3, 1, 0, 128, 0, 0, 1},
    // ---- level 1: 32x32, C=640
    {"L1 conv3x3 640>640", 32, 32, 32, 640, 0, 3, 1, 0, 640, 0, 7},
    {"L1 conv3x3 1280cat>640", 32, 32, 32, 640, 640, 3, 1, 0, 640, 0, 2},
    {"L1 lin 640>640 +res", 32, 32, 32, 640, 0, 1, 1, 0, 640, 3, 40},
    {"L1 qkv 640>1920 split", 32, 32, 32, 640, 0, 1, 1, 0, 1920, 2, 15},
    {"L1 geglu 640>5120", 32, 32, 32, 640, 0, 1, 1, 0, 5120, 1, 10},
    {"L1 ffout 2560>640 +res", 32, 32, 32, 2560, 0, 1, 1, 0, 640, 3, 10},
    {"L1 up2x 640>640", 32, 32, 32, 640, 0, 3, 1, 1, 640, 0, 1},
    // ---- level 2: 16x16, C=1280
    {"L2 conv3x3 1280>1280", 32, 16, 16, 1280, 0, 3, 1, 0, 1280, 0, 7},
    {"L2 conv3x3 2560cat>1280", 32, 16, 16, 1280, 1280, 3, 1, 0, 1280, 0, 2},
    {"L2 lin 1280>1280 +res", 32, 16, 16, 1280, 0, 1, 1, 0, 1280, 3, 40},
    {"L2 geglu 1280>10240", 32, 16, 16, 1280, 0, 1, 1, 0, 10240, 1, 10},
    {"L2 ffout 5120>1280 +res", 32, 16, 16, 5120, 0, 1, 1, 0, 1280, 3, 10},
    // ---- level 3: 8x8, C=1280
    {"L3 conv3x3 1280>1280", 32, 8, 8, 1280, 0, 3, 1, 0, 1280, 0, 8},
    {"L3 conv3x3 2560cat>1280", 32, 8, 8, 1280, 1280, 3, 1, 0, 1280, 0, 3},
    {"L3 lin 1280>1280 +res", 32, 8, 8, 1280, 0, 1, 1, 0, 1280, 3, 8},
    // ---- small-channel convs
    {"conv_in 8>320", 32, 64, 64, 8, 0, 3, 1, 0, 320, 0, 1},
    {"conv_out 320>8", 32, 64, 64, 320, 0, 3, 1, 0, 8, 0, 1},
    // ---- VAE decode, 4-frame chunk
    {"VAE 64^2 512>512", 4, 64, 64, 512, 0, 3, 1, 0, 512, 0, 0},
    {"VAE 256^2 256>256", 4, 256, 256, 256, 0, 3, 1, 0, 256, 0, 0},
    {"VAE 512^2 128>128", 4, 512, 512, 128, 0, 3, 1, 0, 128, 0, 0},
    {"VAE up 256>512^2 256>256", 4, 256, 256, 256, 0, 3, 1, 1, 256, 0, 0},
};

typedef int (*gemm_fn)(const vx_gemm_params*, void*);
typedef int (*ln_fn)(const void*, int, int, int, float, const float*, const float*, const float*, int, int, void*, int,
                     void*);
typedef int (*gn_fn)(const void*, int, const void*, int, int, int, int, float, const float*, const float*, int, void*,
                     float*, int, int, int, void*);

// HBM-bound kernels: LayerNorm / GroupNorm at the UNet sizes; GB/s = algorithmic bytes (LN: read+write; GN: 2 reads +
// 1 write) / time
static void norm_bench(void* lib, hipStream_t st, int reps) {
  ln_fn ln = (ln_fn)dlsym(lib, "vx_layernorm");
  gn_fn gn = (gn_fn)dlsym(lib, "vx_groupnorm");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct { const char* name; int frames, hw, c; } cases[] = {
      {"L0 320", 32, 4096, 320}, {"L1 640", 32, 1024, 640}, {"L2 1280", 32, 256, 1280}, {"L3 1280", 32, 64, 1280},
      {"L0 cat 640", 32, 4096, 640}, {"VAE 512^2 128 x4", 4, 262144, 128}};
  float *gamma, *beta, *ws;
  CK(hipMalloc(&gamma, 4096 * 4)); CK(hipMalloc(&beta, 4096 * 4)); CK(hipMalloc(&ws, 64 << 20));
  fill_f32<<<16, 256, 0, st>>>(gamma, 4096, 7u, 1.0f);
  fill_f32<<<16, 256, 0, st>>>(beta, 4096, 8u, 1.0f);
  printf("%-22s %10s %10s %10s %10s\n", "norm case", "LN us", "LN GB/s", "GN us", "GN GB/s");
  for (auto& c : cases) {
    size_t n = (size_t)c.frames * c.hw * c.c;
    bf16_t *x, *y;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2));
    fill_bf16<<<2048, 256, 0, st>>>(x, n, 1u, 1.0f);
    int rows = c.frames * c.hw;
    float ms_ln = 0, ms_gn = 0;
    if (c.c <= 2048 && c.c >= 320) {
      for (int i = 0; i < 2; ++i) ln(x, c.c, rows, c.c, 1e-5f, gamma, beta, nullptr, 1, 1, y, c.c, st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) ln(x, c.c, rows, c.c, 1e-5f, gamma, beta, nullptr, 1, 1, y, c.c, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_ln, e0, e1));
    }
    int slices = c.hw / 16 < 1 ? 1 : (c.hw / 16 > 64 ? 64 : c.hw / 16);
    for (int i = 0; i < 2; ++i) gn(x, c.c, nullptr, 0, c.frames, c.hw, 32, 1e-5f, gamma, beta, 1, y, ws, slices, c.hw, 0, st);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) gn(x, c.c, nullptr, 0, c.frames, c.hw, 32, 1e-5f, gamma, beta, 1, y, ws, slices, c.hw, 0, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_gn, e0, e1));
    double us_ln = 1e3 * ms_ln / reps, us_gn = 1e3 * ms_gn / reps;
    printf("%-22s %10.1f %10.0f %10.1f %10.0f\n", c.name, us_ln, us_ln > 0 ? 4.0 * n / us_ln * 1e-3 : 0.0, us_gn,
           6.0 * n / us_gn * 1e-3);
    CK(hipFree(x)); CK(hipFree(y));
  }
}

typedef const char* (*err_fn)(void);

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <libvexpress_hip.so> [reps=5] [name filter]\n", argv[0]);
    return 1;
  }
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  gemm_fn gemm = (gemm_fn)dlsym(lib, "vx_gemm");
  err_fn lasterr = (err_fn)dlsym(lib, "vx_last_error_string");
  // optional ablation build (make -C v-express_amd/csrc ablate): ABLATE=0,1,3,4,8,16 times every shape once per flag set
  typedef int (*abl_fn)(int);
  abl_fn set_ablate = (abl_fn)dlsym(lib, "vx_gemm_set_ablate");
  std::vector<int> ablate_list = {0};
  if (set_ablate && getenv("ABLATE")) {
    ablate_list.clear();
    std::string e = getenv("ABLATE");
    size_t pos = 0;
    while (pos < e.size()) {
      ablate_list.push_back(atoi(e.c_str() + pos));
      size_t c = e.find(',', pos);
      if (c == std::string::npos) break;
      pos = c + 1;
    }
  }
  // RING_TRACE=1 (ablation build): dump the slot timestamps of waves 0 / 4 of block 0 of the ring kernel
  typedef int (*trace_fn)(void*);
  trace_fn set_trace = (trace_fn)dlsym(lib, "vx_gemm_ring_set_trace");
  unsigned long long* d_trace = nullptr;
  if (set_trace && getenv("RING_TRACE")) CK(hipMalloc(&d_trace, 2 * 512 * 8));
  int reps = argc > 2 ? atoi(argv[2]) : 5;
  const char* filter = argc > 3 ? argv[3] : nullptr;
  const int NS = 4096;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int *d_sm, *d_sn;
  float *d_ref, *d_got;
  CK(hipMalloc(&d_sm, NS * 4)); CK(hipMalloc(&d_sn, NS * 4));
  CK(hipMalloc(&d_ref, NS * 4)); CK(hipMalloc(&d_got, NS * 4));
  std::vector<int> sm(NS), sn(NS);
  std::vector<float> ref(NS), got(NS);
  double tot_us = 0, tot_fl = 0;
  printf("%-30s %8s %6s %6s %9s %8s %9s %s\n", "shape", "M", "N", "K", "us", "TF/s", "maxrelerr", "ok");
  for (const Shape& s : SHAPES) {
    if (filter && !strstr(s.name, filter)) continue;
    int pad = s.prepad ? 0 : s.ks / 2;
    int he = s.h << s.up, we = s.w << s.up;
    int ho = (he + 2 * pad - s.ks) / s.stride + 1, wo = (we + 2 * pad - s.ks) / s.stride + 1;
    int cin = s.c1 + s.c2, K = s.ks * s.ks * cin, M = s.nb * ho * wo, N = s.n;
    size_t rows_in = (size_t)s.nb * s.h * s.w;
    bf16_t *a, *a2 = nullptr, *w, *out, *res = nullptr, *p1 = nullptr, *p2 = nullptr;
    float* bias;
    CK(hipMalloc(&a, rows_in * s.c1 * 2));
    fill_bf16<<<2048, 256, 0, st>>>(a, rows_in * s.c1, 1u, 1.0f);
    if (s.c2) {
      CK(hipMalloc(&a2, rows_in * s.c2 * 2));
      fill_bf16<<<2048, 256, 0, st>>>(a2, rows_in * s.c2, 2u, 1.0f);
    }
    CK(hipMalloc(&w, (size_t)N * K * 2));
    fill_bf16<<<2048, 256, 0, st>>>(w, (size_t)N * K, 3u, 1.0f / sqrtf((float)K));
    CK(hipMalloc(&bias, N * 4));
    fill_f32<<<(N + 255) / 256, 256, 0, st>>>(bias, N, 4u, 0.5f);
    int nout = s.epi == 1 ? N / 2 : N;
    CK(hipMalloc(&out, (size_t)M * nout * 2));
    vx_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.a2 = a2; p.c1 = s.c1; p.c2 = s.c2; p.lda1 = s.c1; p.lda2 = s.c2;
    p.nb = s.nb; p.h_in = s.h; p.w_in = s.w; p.kh = p.kw = s.ks; p.stride = s.stride; p.pad = pad;
    p.upsample = s.up; p.h_out = ho; p.w_out = wo;
    p.w = w; p.n = N; p.k = K; p.m = M;
    p.alpha = 1.0f; p.bias = bias;
    p.out = out; p.ldc = nout;
    if (s.epi == 1) p.epi = VX_EPI_GEGLU;
    else if (s.epi == 2) {
      p.epi = VX_EPI_SPLIT;
      int c = N / 3, hw = ho * wo;
      CK(hipMalloc(&p1, (size_t)M * c * 2));
      CK(hipMalloc(&p2, (size_t)M * c * 2));
      p.part_cols = c; p.n_parts = 3;
      p.part_out[0] = out; p.part_out[1] = p1; p.part_out[2] = p2;
      p.part_kind[0] = p.part_kind[1] = VX_PART_ROWS; p.part_kind[2] = VX_PART_VT;
      p.part_ld[0] = p.part_ld[1] = c;
      p.seq_len = hw; p.head_dim = c / 8; p.vt_pitch = hw;
      p.ldc = c;
    } else {
      p.epi = VX_EPI_STORE;
      if (s.epi == 3) {
        CK(hipMalloc(&res, (size_t)M * N * 2));
        fill_bf16<<<2048, 256, 0, st>>>(res, (size_t)M * N, 5u, 1.0f);
        p.residual = res; p.ldr = N;
      }
    }
    // LNFOLD=1: the product form of the q / qkv / GEGLU projections (vx_gemm_params.ln_stats: timing only, the spot
    // check below is skipped for these launches)
    float *lnst = nullptr, *lncs = nullptr;
    const bool lnfold = getenv("LNFOLD") && atoi(getenv("LNFOLD")) && s.ks == 1 && s.epi != 3;
    if (lnfold) {
      CK(hipMalloc(&lnst, (size_t)M * 2 * 4));
      CK(hipMalloc(&lncs, (size_t)N * 4));
      fill_f32<<<(2 * M + 255) / 256, 256, 0, st>>>(lnst, (size_t)2 * M, 11u, 0.25f);
      fill_f32<<<(N + 255) / 256, 256, 0, st>>>(lncs, N, 12u, 0.5f);
      p.ln_stats = lnst; p.ln_colsum = lncs;
    }
    if (getenv("RING_HINT")) p.ring_hint = atoi(getenv("RING_HINT"));   // 1: the persistent kernel whenever structurally eligible
    void* skws = nullptr;
    if (getenv("SPLITK") && atoi(getenv("SPLITK")) > 1 && p.epi == VX_EPI_STORE && K / 64 >= atoi(getenv("SPLITK"))) {
      p.splitk = atoi(getenv("SPLITK"));   // two-launch deterministic split-K (the 8x8-level policy of ops._splitk)
      CK(hipMalloc(&skws, (size_t)p.splitk * M * N * 4));
      p.splitk_ws = skws;
    }
    int rc = gemm(&p, st);
    if (rc != 0) {
      printf("%-30s launch error %d: %s\n", s.name, rc, lasterr());
      continue;
    }
    CK(hipStreamSynchronize(st));
    // ---- spot check (STORE and the Q part of SPLIT are directly comparable)
    double maxrel = -1;
    bool ok = true;
    if (!lnfold) {
      int ncheck = s.epi == 2 ? N / 3 : (s.epi == 1 ? N / 2 : N);
      for (int i = 0; i < NS; ++i) {
        uint32_t h = hash32(i * 7919u + 17u);
        // bias the samples towards tile / image borders
        int m = (i % 4 == 0) ? (int)(h % 64) * (M / 64) + (int)((h >> 8) % 3) - 1 : (int)(h % (uint32_t)M);
        if (m < 0) m = 0;
        if (m >= M) m = M - 1;
        sm[i] = m;
        sn[i] = (int)(hash32(h) % (uint32_t)ncheck);
      }
      CK(hipMemcpy(d_sm, sm.data(), NS * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_sn, sn.data(), NS * 4, hipMemcpyHostToDevice));
      RefArgs r = {a, a2, w, bias, res, s.c1, s.c2, s.c1, s.c2, s.nb, s.h, s.w, s.ks, s.ks, s.stride, pad, s.up,
                   ho, wo, N, K, M, N, s.epi == 1 ? 1 : 0};
      ref_samples<<<NS / 64, 64, 0, st>>>(r, d_sm, d_sn, d_ref, NS);
      gather_out<<<NS / 64, 64, 0, st>>>(out, p.ldc, d_sm, d_sn, d_got, NS);
      CK(hipMemcpyAsync(ref.data(), d_ref, NS * 4, hipMemcpyDeviceToHost, st));
      CK(hipMemcpyAsync(got.data(), d_got, NS * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      double scale = 0;
      for (int i = 0; i < NS; ++i) scale = fmax(scale, fabs(ref[i]));
      maxrel = 0;
      for (int i = 0; i < NS; ++i) maxrel = fmax(maxrel, fabs(ref[i] - got[i]) / (scale + 1e-20));
      ok = maxrel < 1.0 / 128;   // bf16 output rounding: 2^-8 relative to the value, <= 2^-7 of max
    }
    double us = 0, fl = 2.0 * M * N * K;
    for (size_t ai = 0; ai < ablate_list.size(); ++ai) {
      if (set_ablate) set_ablate(ablate_list[ai]);
      for (int i = 0; i < 2; ++i) gemm(&p, st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) gemm(&p, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double u = 1e3 * ms / reps;
      if (ai == 0) {
        us = u;
        printf("%-30s %8d %6d %6d %9.1f %8.1f %9.2e %s\n", s.name, M, N, K, us, fl / us * 1e-6, maxrel,
               ok ? "ok" : "MISMATCH");
      } else {
        printf("    ablate=%-3d %9.1f us\n", ablate_list[ai], u);
      }
    }
    if (set_ablate) set_ablate(0);
    if (d_trace) {
      CK(hipMemset(d_trace, 0, 2 * 512 * 8));
      set_trace(d_trace);
      gemm(&p, st);
      CK(hipStreamSynchronize(st));
      set_trace(nullptr);
      std::vector<unsigned long long> tr(1024);
      CK(hipMemcpy(tr.data(), d_trace, 1024 * 8, hipMemcpyDeviceToHost));
      if (tr[0]) {
        // 7 stamps per phase: reads issued, DMA issued, vmcnt waited, lgkmcnt waited, barrier, MFMAs done, barrier
        for (int g = 0; g < 2; ++g) {
          printf("  trace wave %d (ticks per phase: rd|dma|vmwait|lgkm|bar|M|bar):\n   ", g * 4);
          for (int i = 7; i + 6 < 512 && tr[g * 512 + i + 6]; i += 7) {
            unsigned long long* q = &tr[g * 512 + i];
            printf(" [%ld|%ld|%ld|%ld|%ld|%ld|%ld]", (long)(q[0] - q[-1]), (long)(q[1] - q[0]), (long)(q[2] - q[1]),
                   (long)(q[3] - q[2]), (long)(q[4] - q[3]), (long)(q[5] - q[4]), (long)(q[6] - q[5]));
            if ((i / 7) % 4 == 0) printf("\n   ");
            if (i >= 7 * 40) break;
          }
          printf("\n");
        }
      }
    }
    fflush(stdout);
    tot_us += us * s.per_fwd;
    tot_fl += fl * s.per_fwd;
    for (void* q : {(void*)a, (void*)a2, (void*)w, (void*)bias, (void*)out, (void*)res, (void*)p1, (void*)p2, (void*)lnst, (void*)lncs})
      if (q) CK(hipFree(q));
  }
  if (!filter || strstr("norm", filter)) norm_bench(lib, st, reps);
  if (tot_us > 0)
    printf("weighted (per_fwd counts): %.1f ms, %.1f TF/s\n", tot_us * 1e-3, tot_fl / tot_us * 1e-6);
  return 0;
}
