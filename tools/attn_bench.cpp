// Stand-alone benchmark + spot check of vx_attention (no torch).
//   hipcc --offload-arch=gfx950 -O2 -o tools/attn_bench tools/attn_bench.cpp -ldl
//   tools/attn_bench v-express_amd/libvexpress_hip.so [reps]
// Shapes: the spatial self / reference attention of the UNet3D CFG forward at 512^2, f = 16.
// Check: 64 sampled (batch, head, query) rows against an fp64 host softmax over all keys.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

typedef uint16_t bf16_t;
static inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
static inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

typedef int (*attn_fn)(const void*, int, const void*, int, const void*, int, void*, int, int, int, int, int, int, int,
                       float, void*);
typedef const char* (*err_fn)(void);
typedef int (*knorm_fn)(const void*, int, int, int, int, int, float*, void*);
typedef int (*attnb_fn)(const void*, int, const void*, int, const void*, int, void*, int, int, int, int, int, int, int,
                        float, const float*, void*);

struct Shape { const char* name; int batch, heads, nq, nkv, d, q_per_kv, per_fwd; };

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <libvexpress_hip.so> [reps=10] [filter]\n", argv[0]);
    return 1;
  }
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  attn_fn attn = (attn_fn)dlsym(lib, "vx_attention");
  err_fn lasterr = (err_fn)dlsym(lib, "vx_last_error_string");
  // ATTN_BOUND=1: time vx_key_norm_max + vx_attention_bounded (the pair the model issues) instead of vx_attention
  knorm_fn knorm = (knorm_fn)dlsym(lib, "vx_key_norm_max");
  attnb_fn attnb = (attnb_fn)dlsym(lib, "vx_attention_bounded");
  const bool bound = getenv("ATTN_BOUND") && atoi(getenv("ATTN_BOUND")) && knorm && attnb;
  float kscale = getenv("ATTN_KSCALE") ? (float)atof(getenv("ATTN_KSCALE")) : 1.0f;   // key magnitude (looser bound)
  printf("mode: %s, key scale %.1f\n", bound ? "bounded softmax (key norm + attention timed together)" : "exact", kscale);
  int reps = argc > 2 ? atoi(argv[2]) : 10;
  const char* filter = argc > 3 ? argv[3] : nullptr;
  const Shape shapes[] = {
      {"L0 self  32x8 N=4096 d=40", 32, 8, 4096, 4096, 40, 1, 5},
      {"L0 ref   16x8 N=4096 d=40 (shared K/V)", 16, 8, 4096, 4096, 40, 16, 5},
      {"L1 self  32x8 N=1024 d=80", 32, 8, 1024, 1024, 80, 1, 5},
      {"L1 ref   16x8 N=1024 d=80 (shared K/V)", 16, 8, 1024, 1024, 80, 16, 5},
      {"L2 self  32x8 N=256 d=160", 32, 8, 256, 256, 160, 1, 6},
      {"VAE mid  4x1 N=4096 d=512", 4, 1, 4096, 4096, 512, 1, 0},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-42s %10s %9s %10s %s\n", "shape", "us", "TF/s", "max err", "ok");
  double tot_us = 0;
  for (const Shape& s : shapes) {
    if (filter && !strstr(s.name, filter)) continue;
    const int C = s.heads * s.d, kvb = s.batch / s.q_per_kv;
    const size_t nq = (size_t)s.batch * s.nq * C, nk = (size_t)kvb * s.nkv * C;
    const int pitch = (s.nkv + 7) / 8 * 8;
    const size_t nvt = (size_t)kvb * s.heads * s.d * pitch;
    std::vector<bf16_t> hq(nq), hk(nk), hvt(nvt, 0);
    for (size_t i = 0; i < nq; ++i) hq[i] = f2bf(((int)(hash32((uint32_t)i * 2654435761u + 1u) & 0xffff) - 32768) / 16384.0f);
    for (size_t i = 0; i < nk; ++i)
      hk[i] = f2bf(kscale * ((int)(hash32((uint32_t)i * 2654435761u + 2u) & 0xffff) - 32768) / 16384.0f);
    for (int b = 0; b < kvb; ++b)
      for (int h = 0; h < s.heads; ++h)
        for (int dd = 0; dd < s.d; ++dd)
          for (int t = 0; t < s.nkv; ++t)
            hvt[(((size_t)b * s.heads + h) * s.d + dd) * pitch + t] =
                f2bf(((int)(hash32((uint32_t)((((size_t)b * s.nkv + t) * s.heads + h) * s.d + dd) + 3u) & 0xffff) - 32768) / 32768.0f);
    bf16_t *dq, *dk, *dvt, *dout;
    CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dvt, nvt * 2)); CK(hipMalloc(&dout, nq * 2));
    CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvt, hvt.data(), nvt * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, nq * 2));
    // ATTN_PRESCALED=1: the keys carry scale * log2(e) (what the model folds into to_k) and the kernels get scale = 0
    const bool pre = getenv("ATTN_PRESCALED") && atoi(getenv("ATTN_PRESCALED"));
    const float scale_true = 1.0f / sqrtf((float)s.d);
    if (pre)
      for (size_t i = 0; i < nk; ++i) hk[i] = f2bf(bf2f(hk[i]) * scale_true * 1.4426950408889634f);
    if (pre) CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
    const float scale = pre ? 0.0f : scale_true;
    const double scale_ref = pre ? 0.6931471805599453 : (double)scale_true;
    float* dkm = nullptr;
    CK(hipMalloc(&dkm, sizeof(float) * kvb * s.heads));
    auto run = [&]() -> int {
      if (!bound) return attn(dq, C, dk, C, dvt, pitch, dout, C, s.batch, s.heads, s.nq, s.nkv, s.d, s.q_per_kv, scale, st);
      int r = knorm(dk, C, kvb, s.heads, s.nkv, s.d, dkm, st);
      if (r) return r;
      return attnb(dq, C, dk, C, dvt, pitch, dout, C, s.batch, s.heads, s.nq, s.nkv, s.d, s.q_per_kv, scale, dkm, st);
    };
    int rc = run();
    if (rc != 0) {
      printf("%-42s launch error %d: %s\n", s.name, rc, lasterr());
      continue;
    }
    CK(hipStreamSynchronize(st));
    std::vector<bf16_t> ho(nq);
    CK(hipMemcpy(ho.data(), dout, nq * 2, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int smp = 0; smp < 64; ++smp) {
      uint32_t hsh = hash32(smp * 7919u + 11u);
      const int b = hsh % s.batch, h = (hsh >> 8) % s.heads;
      const int qi = smp < 8 ? (smp < 4 ? smp : s.nq - 1 - (smp - 4)) : (int)((hsh >> 12) % s.nq);
      const int kb = b / s.q_per_kv;
      std::vector<double> sc(s.nkv);
      double mx = -1e300;
      for (int t = 0; t < s.nkv; ++t) {
        double a = 0;
        for (int dd = 0; dd < s.d; ++dd)
          a += (double)bf2f(hq[((size_t)b * s.nq + qi) * C + h * s.d + dd]) *
               (double)bf2f(hk[((size_t)kb * s.nkv + t) * C + h * s.d + dd]);
        sc[t] = a * scale_ref;
        mx = fmax(mx, sc[t]);
      }
      double den = 0;
      for (int t = 0; t < s.nkv; ++t) { sc[t] = exp(sc[t] - mx); den += sc[t]; }
      for (int dd = 0; dd < s.d; ++dd) {
        double o = 0;
        for (int t = 0; t < s.nkv; ++t)
          o += sc[t] * (double)bf2f(hvt[(((size_t)kb * s.heads + h) * s.d + dd) * pitch + t]);
        o /= den;
        double got = bf2f(ho[((size_t)b * s.nq + qi) * C + h * s.d + dd]);
        maxerr = fmax(maxerr, fabs(got - o));
        maxref = fmax(maxref, fabs(o));
      }
    }
    bool ok = maxerr <= maxref / 64.0 + 1e-4;
    for (int i = 0; i < 2; ++i) run();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = 1e3 * ms / reps, fl = 4.0 * s.batch * s.heads * (double)s.nq * s.nkv * s.d;
    printf("%-42s %10.1f %9.1f %10.3e %s (max|ref| %.3f)\n", s.name, us, fl / us * 1e-6, maxerr, ok ? "ok" : "MISMATCH", maxref);
    fflush(stdout);
    tot_us += us * s.per_fwd;
    CK(hipFree(dq)); CK(hipFree(dk)); CK(hipFree(dvt)); CK(hipFree(dout)); CK(hipFree(dkm));
  }
  printf("per CFG forward (self + reference attention, counts per_fwd): %.2f ms\n", tot_us * 1e-3);
  return 0;
}
