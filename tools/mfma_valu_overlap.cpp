// Can the VALU (v_exp_f32 / v_fma_f32) of a gfx950 SIMD run while its matrix core executes v_mfma_f32_16x16x32_bf16?
// Per iteration a wave issues 8 independent MFMAs (4 accumulators x 2), and/or 16 v_exp_f32, and/or 16 v_fma_f32,
// interleaved in one instruction stream; 1, 2 and 3 waves per SIMD.  If the combined time is ~ max(parts) the pipes
// overlap; if it is ~ the sum they do not (then a flash-attention wave's softmax cannot hide under its MFMAs).
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_valu_overlap tools/mfma_valu_overlap.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool MFMA, bool EXP, bool FMA>
__global__ void k(float* out, int iters, float seed) {
  f32x4_t acc[4];
  float e[16];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{seed, seed, seed, seed};
  for (int i = 0; i < 16; ++i) e[i] = seed * 1e-3f + threadIdx.x * 1e-6f + i * 1e-4f;
  uint4 au = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u), bu = au;
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, au), b = __builtin_bit_cast(bf16x8_t, bu);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MFMA) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
      if (EXP) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * j]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * j + 1]));
      }
      if (FMA) {
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[(2 * j + 8) & 15]));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[(2 * j + 9) & 15]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += e[i];
  if (s == 12345.f) out[0] = s;
}

template <bool MFMA, bool EXP, bool FMA>
void run(const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, 4));
  for (int wps : {1, 2, 3}) {
    const int iters = 20000, threads = 256 * wps, blocks = 256;
    k<MFMA, EXP, FMA><<<blocks, threads>>>(out, 100, 1.f);
    CK(hipEventRecord(e0));
    k<MFMA, EXP, FMA><<<blocks, threads>>>(out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s waves/SIMD %d: %7.1f ns per iteration per wave-slot, %7.1f ns per iteration per SIMD\n", name, wps,
           ms * 1e6 / iters, ms * 1e6 / iters / wps);
  }
}
int main() {
  run<true, false, false>("8 mfma");
  run<false, true, false>("16 v_exp");
  run<false, false, true>("16 v_fma");
  run<true, true, false>("8 mfma + 16 v_exp");
  run<true, false, true>("8 mfma + 16 v_fma");
  run<true, true, true>("8 mfma + 16 v_exp + 16 v_fma");
  run<false, true, true>("16 v_exp + 16 v_fma");
  return 0;
}
