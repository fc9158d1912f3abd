// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libvexpress_hip.so.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (A: lane l holds row l&15, k = 8*(l>>4)..+7;
// B: col l&15, same k; C/D: col = l&15, row = 4*(l>>4)+reg).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// ---- element type of the build.  ONE set of sources, two libraries: libvexpress_hip.so computes on bfloat16 storage,
// libvexpress_hip_f16.so (-DVX_ELEM_F16) on IEEE half - the reference's own default (inference.py:44,150-151 `--dtype fp16`);
// both accumulate in fp32.  `bf16_t` = the raw 16 bits of an element of the build (the name predates the second library);
// every conversion, literal and matrix instruction that depends on the format goes through the helpers below.
typedef uint16_t bf16_t;
#ifdef VX_ELEM_F16
#define VX_ELEM_NAME "f16"
typedef _Float16 vx_e16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 vx_e16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 vx_e16x2_t __attribute__((ext_vector_type(2)));
#define VX_E16_ONE 0x3c00u          // 1.0
#define VX_E16_ONE2 0x3c003c00u
#define VX_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define VX_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define VX_MFMA_16x16x16 __builtin_amdgcn_mfma_f32_16x16x16f16
#else
#define VX_ELEM_NAME "bf16"
typedef __bf16 vx_e16x8_t __attribute__((ext_vector_type(8)));
typedef short vx_e16x4_t __attribute__((ext_vector_type(4)));     // (the 16x16x16 bf16_1k builtin takes shorts)
typedef __bf16 vx_e16x2_t __attribute__((ext_vector_type(2)));
#define VX_E16_ONE 0x3f80u
#define VX_E16_ONE2 0x3f803f80u
#define VX_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define VX_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define VX_MFMA_16x16x16 __builtin_amdgcn_mfma_f32_16x16x16bf16_1k
#endif
typedef vx_e16x8_t bf16x8_t;
// Bounded softmax (vx_attention_bounded: the shift is the Cauchy-Schwarz bound m >= max_j s_ij instead of the row maximum, so
// every p = 2^(s - m) <= 1 and rows far under their bound are small numbers).  bfloat16 carries p down to 2^-126; IEEE half
// does not: its build lowers the shift by VX_P_HEADROOM log2 units (p <= 2^14 < 65504; the factor is per row and cancels in
// the normalisation) and sends a row to the exact recompute (true row maximum as the shift) as soon as rounding / flushing
// its probabilities below half's normal range could cost more than half's own epsilon: absolute error <= n_kv 2^-25 against
// a row sum l, i.e. l < n_kv 2^-14.  (bf16 build: l < 2^-100, as before.)
#ifdef VX_ELEM_F16
#define VX_P_HEADROOM 14.0f
#define VX_P_MIN_ROWSUM(n_kv) ((float)(n_kv) * 6.103515625e-5f)
#else
#define VX_P_HEADROOM 0.0f
#define VX_P_MIN_ROWSUM(n_kv) 7.8886e-31f   // 2^-100
#endif

#define VX_OK 0
#define VX_ERR_INVALID (-1)
#define VX_ERR_UNSUPPORTED (-2)
#define VX_ERR_HIP (-3)

extern "C" void vx_set_error(const char* fmt, ...);
int vx_check_launch(const char* what);
// name of the kernel instantiation the last launch of this thread's MFMA entry points (vx_gemm, vx_attention*,
// vx_temporal_attention, vx_ff_fused, vx_tblock_fused) made, spelled as rocprofv3 prints it (vx_last_kernel): set by the
// launch templates themselves, so it is exact by construction
extern thread_local const char* g_vx_last_kernel;

#define VX_REQUIRE(cond, ...)                   \
  do {                                          \
    if (!(cond)) {                              \
      vx_set_error(__VA_ARGS__);                \
      return VX_ERR_INVALID;                    \
    }                                           \
  } while (0)

typedef float vx_f32x2_t __attribute__((ext_vector_type(2)));
#ifdef VX_ELEM_F16
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even (v_cvt_f16_f32), NaN preserved, |f| > 65504 -> inf
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
// the two elements of a packed pair (v_cvt_f32_f16, the high half through SDWA)
__device__ __forceinline__ float e16_lo(uint32_t u) { return (float)__builtin_bit_cast(vx_e16x2_t, u)[0]; }
__device__ __forceinline__ float e16_hi(uint32_t u) { return (float)__builtin_bit_cast(vx_e16x2_t, u)[1]; }
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float e16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float e16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#endif

// two floats -> packed element pair with one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (round-to-nearest-even, NaN preserved)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  vx_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vx_e16x2_t));
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = e16_lo(v.x); f[1] = e16_hi(v.x);
  f[2] = e16_lo(v.y); f[3] = e16_hi(v.y);
  f[4] = e16_lo(v.z); f[5] = e16_hi(v.z);
  f[6] = e16_lo(v.w); f[7] = e16_hi(v.w);
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf GELU (torch.nn.functional.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below
// the bf16 output rounding): 1 v_rcp + 1 v_exp + ~12 FMA-class ops instead of the ~50-instruction branchy erff().
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-poly, e, 1.0f), x);
}
// erf GELU as relu(x) - |x| T(|x|) with the Gaussian tail T(u) = 0.5 erfc(u / sqrt 2) = 2^q(u): the base-2 logarithm of
// the tail is smooth (q ~ -1 - 1.15 u - 0.72 u^2 ...), so a degree-6 polynomial (weighted minimax fit on [0, 8], weight
// u T(u): the sensitivity of the RESULT) reproduces x Phi(x) to |abs err| <= 5.1e-7 in float32 over the whole line and to
// <= 4e-4 relative for |x| < 4 (bf16 output rounding: 4e-3) - the accuracy class of the Abramowitz-Stegun form above
// (1.5e-7 |x|) with 6 FMA + 1 v_exp_f32 + 3 plain ops instead of ~14 plain ops + v_rcp_f32 + v_exp_f32.  Round 3: the
// GEGLU epilogue of a K = 320 GEMM is VALU-bound (80 GELUs per thread and tile against 100 MFMAs), so the op count of
// this function is launch time.  |x| is clamped at 8 (T(8) = 6e-16; keeps inf finite inside the polynomial).
__device__ __forceinline__ float gelu_f(float x) {
  // v_med3_f32 for the clamp and the relu: fminf / fmaxf cost an extra canonicalising v_max each under IEEE mode
  const float u = __builtin_amdgcn_fmed3f(fabsf(x), 0.0f, 8.0f);
  float q = fmaf(3.309281237e-05f, u, -7.692196523e-04f);
  q = fmaf(q, u, 8.080716245e-03f);
  q = fmaf(q, u, -5.341210216e-02f);
  q = fmaf(q, u, -4.587709606e-01f);
  q = fmaf(q, u, -1.151201725e+00f);
  q = fmaf(q, u, -9.999930859e-01f);
#ifdef VX_GELU_NOASM   // A/B build: the compiler's relu (v_max + a canonicalising v_max under IEEE mode)
  const float relu = fmaxf(x, 0.0f);
#else
  float relu;   // one v_max_f32
  asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(x));
#endif
  return fmaf(-u, __builtin_amdgcn_exp2f(q), relu);
}
// value * gelu(gate) for four (value, gate) pairs - the inner step of both GEGLU epilogues.
// VX_GELU_PK (A/B build, round 4): the degree-6 polynomial runs as v_pk_fma_f32 over element PAIRS - six packed FMAs per two
// elements instead of six v_fmaak_f32 (8-byte encodings with a 32-bit literal) per element.  The coefficients sit in four
// register pairs that are made opaque to the compiler (it would re-materialise literal pairs in front of every use) and are
// broadcast to both halves by op_sel; same arithmetic (fp32 FMA chain in the same order), same bits.
typedef float vx_f2 __attribute__((ext_vector_type(2)));
#ifdef VX_GELU_PK
struct GeluPk {
  vx_f2 A, B, C, D;   // {c6, c5}, {c4, c3}, {c2, c1}, {c0, -}
  __device__ __forceinline__ void init() {
    A = vx_f2{3.309281237e-05f, -7.692196523e-04f};
    B = vx_f2{8.080716245e-03f, -5.341210216e-02f};
    C = vx_f2{-4.587709606e-01f, -1.151201725e+00f};
    D = vx_f2{-9.999930859e-01f, 0.0f};
    asm volatile("" : "+v"(A), "+v"(B), "+v"(C), "+v"(D));
  }
  __device__ __forceinline__ void mul2(float v0, float v1, float g0, float g1, float& o0, float& o1) const {
    const vx_f2 u = {__builtin_amdgcn_fmed3f(fabsf(g0), 0.0f, 8.0f), __builtin_amdgcn_fmed3f(fabsf(g1), 0.0f, 8.0f)};
    vx_f2 q;
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(q) : "v"(A), "v"(u));   // c6 u + c5
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(q) : "v"(u), "v"(B));                  // q u + c4
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(q) : "v"(u), "v"(B));   // q u + c3
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(q) : "v"(u), "v"(C));                  // q u + c2
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(q) : "v"(u), "v"(C));   // q u + c1
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(q) : "v"(u), "v"(D));                  // q u + c0
    float r0, r1;
    asm("v_max_f32 %0, 0, %1" : "=v"(r0) : "v"(g0));
    asm("v_max_f32 %0, 0, %1" : "=v"(r1) : "v"(g1));
    const vx_f2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const vx_f2 gl = vx_f2{r0, r1} - u * e;
    const vx_f2 o = vx_f2{v0, v1} * gl;
    o0 = o.x;
    o1 = o.y;
  }
};
#endif
// the Abramowitz-Stegun form (kept for A/B builds: -DVX_GELU_AS)
__device__ __forceinline__ float gelu_as_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
#ifdef VX_GELU_AS
#define gelu_f gelu_as_f
#endif

__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
  return VX_MFMA_16x16x32(__builtin_bit_cast(vx_e16x8_t, a), __builtin_bit_cast(vx_e16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  return VX_MFMA_32x32x16(__builtin_bit_cast(vx_e16x8_t, a), __builtin_bit_cast(vx_e16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_xor_max(float v, int mask) { return fmaxf(v, __shfl_xor(v, mask, 64)); }
__device__ __forceinline__ float wave_xor_sum(float v, int mask) { return v + __shfl_xor(v, mask, 64); }
// max over the four lanes {l, l^16, l^32, l^48} (one value per 16-lane row) with no LDS round trip: v_permlane16_swap
// exchanges the odd 16-lane rows of its first operand with the even rows of the second, v_permlane32_swap the upper
// half of the first with the lower half of the second.  With the same value in both operands the two results hold
// "my row" and "my partner row" in every lane.  (__shfl_xor compiles to ds_bpermute_b32: ~100 cycles of LDS latency in
// the middle of the online-softmax dependency chain row max -> exp.)
__device__ __forceinline__ float wave_rows_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
