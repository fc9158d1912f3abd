// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction of v_fma_f32 / v_exp_f32 / v_max3_f32 /
// v_cvt_pk_bf16_f32 / v_pk_mul_f32, alone and mixed, at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_bench tools/valu_bench.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (MODE == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
      if (MODE == 4) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[(i + 4) & 7])); }
      if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 6) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if (MODE == 7) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(a[i]));
      if (MODE == 8) asm volatile("v_lshl_add_u32 %0, %0, 23, %0" : "+v"(a[i]));
      // round 4: the GELU polynomial of the GEGLU epilogue is six v_fmaak_f32 (8-byte encoding, 32-bit literal) per
      // element; as v_pk_fma_f32 over element pairs (coefficients in register pairs, broadcast by op_sel) it is three
      if (MODE == 10) asm volatile("v_fmaak_f32 %0, %0, %0, 0x3c0464fb" : "+v"(a[i]));
      if (MODE == 12) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[i]) : "s"(seed));
    }
    if (MODE == 9 || MODE == 11 || MODE == 13) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2* p2 = reinterpret_cast<f2*>(a);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p2[i]));
        if (MODE == 11) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p2[i]));
        if (MODE == 13) asm volatile("v_pk_fma_f32 %0, %0, %0, %1 op_sel_hi:[1,1,0]" : "+v"(p2[i]) : "v"(p2[(i + 1) & 3]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p2[i]));
        if (MODE == 11) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p2[i]));
        if (MODE == 13) asm volatile("v_pk_fma_f32 %0, %0, %0, %1 op_sel_hi:[1,1,0]" : "+v"(p2[i]) : "v"(p2[(i + 1) & 3]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.f) out[0] = s;
}

template <int MODE>
void run(const char* name, int per_iter) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, 4));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  for (int wps : {1, 2, 4}) {
    const int iters = 20000, threads = 256 * wps, blocks = 256;   // 4*wps waves per CU = wps per SIMD
    k<MODE><<<blocks, threads>>>(out, 100, 1.f);
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, threads>>>(out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wps waves x iters x per_iter instructions
    double insts = (double)wps * iters * per_iter;
    double ns_per = ms * 1e6 / insts;
    printf("%-28s waves/SIMD %d: %.2f ns per wave-instruction per SIMD  (= %.1f cycles at %.2f GHz nominal)\n", name, wps, ns_per,
           ns_per * clk_khz * 1e-6, clk_khz * 1e-6);
  }
}
int main() {
  run<0>("v_fma_f32", 8); run<1>("v_exp_f32", 8); run<2>("v_max3_f32", 8); run<3>("v_cvt_pk_bf16_f32", 8);
  run<4>("v_exp_f32 + v_fma_f32 pair", 16); run<5>("v_rcp_f32", 8); run<6>("v_exp_f16", 8); run<7>("v_ldexp_f32", 8);
  run<8>("v_lshl_add_u32", 8);
  run<10>("v_fmaak_f32 (literal)", 8); run<12>("v_fma_f32 (sgpr operand)", 8);
  run<9>("v_pk_fma_f32 (2 fma each)", 8); run<13>("v_pk_fma_f32 op_sel bcast", 8); run<11>("v_pk_mul_f32", 8);
  return 0;
}
