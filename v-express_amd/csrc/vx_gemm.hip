// MFMA GEMM / implicit-GEMM convolution for gfx950 (see include/vexpress_hip.h: vx_gemm).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),   bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Block = WARPS_M x WARPS_N waves (4 or 8), tile BM x BN x 64, every wave owns a 64 x (BN/WARPS_N) sub-tile.
// Both operands are K-contiguous, so a K-tile row is 128 B = eight 16-B chunks.  Staging is asynchronous
// global -> LDS DMA (global_load_lds_dwordx4, no VGPR round trip): each thread issues (BM+BN)/(NT/8) 16-B copies
// per K-tile; the A gather computes (frame, iy, ix, ci) per chunk so 3x3/1x1 convs (stride 1/2, fused nearest-2x
// upsample, channel concat of two sources) and plain linears share one kernel, and every out-of-range chunk
// (conv padding, M/N/K tails) is fetched from a 16-byte zero constant so the copies are branch-free.
// LDS rows are XOR-swizzled at 16-B granularity (slot = chunk ^ ((row>>1)&7)); because the DMA writes
// lane-linearly, the swizzle is applied to the per-lane *global* address.  The ds_read_b128 fragment reads of
// 16 rows x 4 k-groups are then bank-conflict-free.
// Pipeline: STAGES LDS buffers, tiles kt+1 .. kt+STAGES-1 in flight while tile kt is multiplied; one raw
// s_barrier per K-tile and a counted s_waitcnt vmcnt (never 0 in steady state when STAGES > 2).
// Epilogue: the MFMAs produce C^T fragments (4 consecutive output columns per lane), stored straight from registers
// with bias / time-embedding rows / activation / residual fused; only V^T parts are transposed through LDS.
#include "vx_common.h"
#include "vx_gemm_common.h"
#include "../../include/vexpress_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

// Compile-time ablation switches (tools/build_ring_variants.sh builds one library per mask with
// -DVX_GEMM_ABLATE=mask; never defined for the product library):
//   1 no epilogue stores   2 no bias/rowbias/residual loads   4 no DMA inside the K loop   8 no MFMA   16 no LDS reads
//   64 GEGLU without the GELU (value * gate)
#ifdef VX_GEMM_ABLATE
#define ABL(bit) (((VX_GEMM_ABLATE) & (bit)) != 0)
#else
#define ABL(bit) false
#endif

struct RowInfo {
  int pix_base;  // frame * h_in * w_in
  int iy0, ix0;  // oy*stride - pad, ox*stride - pad   (very negative when the row is out of range)
};

template <typename T>
__device__ __forceinline__ T sel3(int i, T a, T b, T c) { return i == 0 ? a : (i == 1 ? b : c); }

// 16 zero bytes: the source of every out-of-range chunk (conv padding, M/N/K tails), so the staging loads are
// branch-free and can all be in flight at once
__device__ __attribute__((aligned(16))) const uint4 g_zero16 = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// FAST = every (row, tap) of the A gather is in range (pad == 0, no upsample: linears and convs over a
// pre-padded input), cin, c1 and K are multiples of 64 and both operands span < 4 GiB.  Then the per-thread part
// of every DMA address is a loop-invariant 32-bit byte offset and the per-tile part is wave-uniform (SGPRs):
// the K loop spends no VALU instructions on addressing (a wave64 VALU op costs 4 cycles = 1/8 of an MFMA).
// F8: both operands are OCP e4m3 bytes (a K-tile row is still 128 B = 128 elements = ONE v_mfma_scale_f32_16x16x128_f8f6f4
// with unit block scales per fragment pair instead of two bf16 MFMAs); the accumulators are multiplied by the row scale of
// A and the row scale of W before the epilogue.  FAST addressing only (plain linears over zero-padded K).
typedef int i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4_t mfma16_f8(const uint4& a_lo, const uint4& a_hi, const uint4& b_lo, const uint4& b_hi,
                                             f32x4_t c) {
  const i32x8_t a = {(int)a_lo.x, (int)a_lo.y, (int)a_lo.z, (int)a_lo.w, (int)a_hi.x, (int)a_hi.y, (int)a_hi.z, (int)a_hi.w};
  const i32x8_t b = {(int)b_lo.x, (int)b_lo.y, (int)b_lo.z, (int)b_lo.w, (int)b_hi.x, (int)b_hi.y, (int)b_hi.z, (int)b_hi.w};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// LNF: a LayerNorm folded into the GEMM (vx_gemm_params.ln_stats) - a template flag, not a run-time branch: the
// transform's live values pushed the 256 x 320 instantiations into 150-200 spilled registers when every kernel carried it.
// GNS (round 4; STORE into bf16, whole tiles only): the epilogue also writes the GroupNorm partial sums of the stored
// values to p.gn_ws (vx_gemm_params.gn_ws; one slab = the 64 rows of a wave) - a template flag for the same reason.
template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int EPI, bool FAST, bool F8 = false, bool LNF = false,
          bool GNS = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void gemm_kernel(const vx_gemm_params p) {
  static_assert(!F8 || FAST, "fp8 operands use the FAST addressing only");
  static_assert(!GNS || (EPI == VX_EPI_STORE && !F8 && !LNF), "GroupNorm partial sums come from the plain STORE epilogue");
  static_assert(!LNF || (FAST && !F8), "a folded LayerNorm sits in front of a plain bf16 linear");
  constexpr int ES = F8 ? 1 : 2;            // bytes per operand element
  constexpr int BKE = 128 / ES;             // elements per K-tile (one 128-byte LDS row)
  constexpr int CE = 16 / ES;               // elements per 16-byte chunk
  constexpr int NTHREADS = 64 * WARPS_M * WARPS_N;
  constexpr int RPP = NTHREADS / 8;   // tile rows staged per pass of the whole block
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int NJ = F8 ? (NI % 2 == 0 ? 2 : 1) : (NI <= 5 ? NI : NI / 2);   // B fragments live at once (fp8: 2 x 16 B each)
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int G = A_IT + B_IT;              // DMA instructions per thread per K-tile
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert(WM == 64, "every wave owns a 64-row slab");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  static_assert(NI % NJ == 0, "fragment grouping");
  static_assert(G * (STAGES - 2) < 64, "vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int n_tiles = (p.n + BN - 1) / BN;
  const int nsplit = (EPI == VX_EPI_STORE && p.splitk > 1) ? p.splitk : 1;
  const int lid0 = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid0 % nsplit, lid = lid0 / nsplit;
  const int tile_m = lid / n_tiles, tile_n = lid - tile_m * n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ A1 = (const bf16_t*)p.a;
  const bf16_t* __restrict__ A2 = (const bf16_t*)p.a2;
  const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
  const int cin = p.c1 + p.c2;
  const int c1 = p.c1;
  const int up = p.upsample;
  const int h_eff = p.h_in << up, w_eff = p.w_in << up;
  const int hw_out = p.h_out * p.w_out;
  const int w_in = p.w_in, kw = p.kw, kh = p.kh;
  const int lda1 = p.lda1, lda2 = p.lda2;
  const bf16_t* const zsrc = reinterpret_cast<const bf16_t*>(&g_zero16);

  // ---- per-thread staging coordinates: LDS slot `s` of row r0 + RPP*i holds K-chunk s ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const int cc = (tid & 7) ^ ((r0 >> 1) & 7);   // K-chunk (8 elements) this thread fetches
  // ---- FAST path state
  uint32_t aoff1[A_IT], aoff2[A_IT], boff[B_IT];
  // this block's K-tile range (the whole K loop unless split-K)
  const int nk_total = (p.k + BKE - 1) / BKE;
  const int kt_begin = (int)((long)split * nk_total / nsplit), kt_end = (int)((long)(split + 1) * nk_total / nsplit);
  int s_kt = kt_begin, s_ci, s_kx, s_ky;   // wave-uniform K position of the next tile to issue
  {
    const int tap0 = (kt_begin * BKE) / cin;
    s_ci = kt_begin * BKE - tap0 * cin;
    s_ky = tap0 / kw;
    s_kx = tap0 - s_ky * kw;
  }
  // ---- general path state
  RowInfo ri[A_IT];
  long wrow[B_IT];   // element offset of weight row n (or -1: beyond N)
  int kg = kt_begin * BK + cc * 8;   // K position of this thread's chunk, tracked incrementally: k = ((ky*kw)+kx)*cin + ci
  int ky = 0, kx = 0, ci = 0;
  const uint32_t lds_wave = lds_addr_of(smem) + (wave * 8) * 128;   // this wave's 8-row (1 KiB) slab within each RPP-row group

  if constexpr (FAST) {
    // rows / weight rows beyond M / N are clamped to the last valid one: their products are never stored
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int m = min(m0 + r0 + RPP * i, p.m - 1);
      int fr = m / hw_out;
      int rem = m - fr * hw_out;
      int oy = rem / p.w_out;
      int ox = rem - oy * p.w_out;
      uint32_t pix = (uint32_t)(fr * p.h_in * p.w_in + oy * p.stride * w_in + ox * p.stride);
      aoff1[i] = (pix * (uint32_t)lda1 + (uint32_t)(cc * CE)) * (uint32_t)ES;
      aoff2[i] = (pix * (uint32_t)lda2 + (uint32_t)(cc * CE)) * (uint32_t)ES;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int n = min(n0 + r0 + RPP * i, p.n - 1);
      boff[i] = ((uint32_t)n * (uint32_t)p.k + (uint32_t)(cc * CE)) * (uint32_t)ES;
    }
  } else {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int m = m0 + r0 + RPP * i;
      if (m < p.m) {
        int fr = m / hw_out;
        int rem = m - fr * hw_out;
        int oy = rem / p.w_out;
        int ox = rem - oy * p.w_out;
        ri[i].pix_base = fr * p.h_in * p.w_in;
        ri[i].iy0 = oy * p.stride - p.pad;
        ri[i].ix0 = ox * p.stride - p.pad;
      } else {
        ri[i].pix_base = 0;
        ri[i].iy0 = -(1 << 28);
        ri[i].ix0 = -(1 << 28);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int n = n0 + r0 + RPP * i;
      wrow[i] = n < p.n ? (long)n * p.k : -1;
    }
    int tap = kg / cin;
    ci = kg - tap * cin;
    ky = tap / kw;
    kx = tap - ky * kw;
  }

  auto issue_tile = [&](int stage) {
    const uint32_t sa = lds_wave + stage * STAGE_BYTES;
    const uint32_t sb = sa + BM * 128;
    if constexpr (FAST) {
      // wave-uniform: source, its row stride and the tap's byte offset
      const bool first = s_ci < c1;
      const char* abase = first ? (const char*)A1 + ((long)s_ci + (long)(s_ky * w_in + s_kx) * lda1) * ES
                                : (const char*)A2 + ((long)(s_ci - c1) + (long)(s_ky * w_in + s_kx) * lda2) * ES;
      const char* bbase = (const char*)Wt + (long)s_kt * 128;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) glds16_s(abase, first ? aoff1[i] : aoff2[i], sa + i * RPP * 128);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) glds16_s(bbase, boff[i], sb + i * RPP * 128);
      ++s_kt;
      s_ci += BKE;
      if (s_ci >= cin) {
        s_ci = 0;
        if (++s_kx == kw) { s_kx = 0; ++s_ky; }
      }
    } else {
      const bool kval = ky < kh;
      const bool first = ci < c1;
      const bf16_t* src = first ? A1 + ci : A2 + (ci - c1);
      const int cs = first ? lda1 : lda2;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        int iy = ri[i].iy0 + ky, ix = ri[i].ix0 + kx;
        bool ok = kval && (unsigned)iy < (unsigned)h_eff && (unsigned)ix < (unsigned)w_eff;
        int pix = ri[i].pix_base + (iy >> up) * w_in + (ix >> up);
        const bf16_t* g = ok ? src + (long)pix * (long)cs : zsrc;
        glds16_v(g, sa + i * RPP * 128);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        bool ok = kval && wrow[i] >= 0;
        const bf16_t* g = ok ? Wt + wrow[i] + kg : zsrc;
        glds16_v(g, sb + i * RPP * 128);
      }
      // advance to the next K-tile
      kg += BK;
      ci += BK;
      while (ci >= cin) {
        ci -= cin;
        if (++kx == kw) { kx = 0; ++ky; }
      }
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = kt_end - kt_begin;
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nk) issue_tile(t);

  const int frow = lane & 15, fgrp = lane >> 4;
  int stage = 0;                 // kt % STAGES
  int fill = STAGES - 1;         // (kt + STAGES - 1) % STAGES: the stage freed by iteration kt-1
  for (int kt = 0; kt < nk; ++kt) {
    // Tile kt has landed once this wave's own DMAs for it are retired (in-order: at most the G*(STAGES-2) copies
    // of the younger tiles may still be pending) and the barrier has collected every wave's.  The same barrier
    // orders everyone's fragment reads of iteration kt-1 before that stage is refilled below.
    if (STAGES > 2 && kt + STAGES - 2 < nk) wait_vmcnt<G * (STAGES - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nk && !ABL(4)) issue_tile(fill);
    const char* sa = smem + stage * STAGE_BYTES;
    const char* sb = sa + BM * 128;
    if constexpr (F8) {
      // lane (row, fgrp) contracts chunks fgrp and 4 + fgrp of the 128-byte row on BOTH operands (the order of the
      // 128 k-values inside the MFMA is free as long as A and B agree): the same two conflict-free ds_read_b128
      // patterns as the bf16 loop's kk = 0 / 1
      uint4 af[MI][2];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        af[i][0] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * WM + i * 16 + frow, fgrp));
        af[i][1] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * WM + i * 16 + frow, 4 + fgrp));
      }
#pragma unroll
      for (int j0 = 0; j0 < NI; j0 += NJ) {
        uint4 bfr[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          bfr[j][0] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * WN + (j0 + j) * 16 + frow, fgrp));
          bfr[j][1] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * WN + (j0 + j) * 16 + frow, 4 + fgrp));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j0 + j] = mfma16_f8(bfr[j][0], bfr[j][1], af[i][0], af[i][1], acc[i][j0 + j]);   // D = C^T fragment
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (ABL(16)) continue;
      uint4 af[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * WM + i * 16 + frow, kk * 4 + fgrp));
#pragma unroll
      for (int j0 = 0; j0 < NI; j0 += NJ) {
        uint4 bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * WN + (j0 + j) * 16 + frow, kk * 4 + fgrp));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (ABL(8)) {
              asm volatile("" ::"v"(bfr[j].x), "v"(af[i].x));
              continue;
            }
            acc[i][j0 + j] = mfma16(bfr[j], af[i], acc[i][j0 + j]);   // D = C^T fragment
          }
      }
    }
    }   // !F8
    stage = stage + 1 == STAGES ? 0 : stage + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
  }

  // ------------------------------------------------------------------ epilogue
  // The MFMAs were issued with the operands swapped (weights as the row operand), so each accumulator fragment is a
  // 16x16 block of C^T: lane l holds output row m = i*16 + (l & 15) and the FOUR CONSECUTIVE columns
  // n = j*16 + 4*(l >> 4) + {0..3}.  STORE / GEGLU / row-major SPLIT parts therefore go straight from registers to
  // global memory as 8-byte (4 x bf16) stores - no LDS round trip, no barrier; bias / time-embedding rows /
  // residual are fetched with the same 4-wide pattern.  Only V^T parts are transposed through LDS.
  const float* __restrict__ bias = p.bias;
  const int lrow = lane & 15, lq = lane >> 4;
  const int wrow0 = m0 + wm * WM, wcol0 = n0 + wn * WN;

  if constexpr (F8) {
    // dequantise: acc[m][n] *= a_scale[m] * w_scale[n] (rows / columns beyond M / N clamped: never stored)
    const float* __restrict__ asc = p.a_scale;
    const float* __restrict__ wsc = p.w_scale;
    float sa_[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) sa_[i] = asc[min(wrow0 + i * 16 + lrow, p.m - 1)];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float4 sw = *reinterpret_cast<const float4*>(wsc + min(wcol0 + j * 16 + lq * 4, p.n - 4));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        acc[i][j][0] *= sa_[i] * sw.x; acc[i][j][1] *= sa_[i] * sw.y;
        acc[i][j][2] *= sa_[i] * sw.z; acc[i][j][3] *= sa_[i] * sw.w;
      }
    }
  }

  if constexpr (LNF) {
    // folded LayerNorm: acc <- rstd[m] * (acc - mean[m] * colsum[n])   (see vx_gemm_params.ln_stats)
    const float2* __restrict__ st = reinterpret_cast<const float2*>(p.ln_stats);
    const float* __restrict__ cs = p.ln_colsum;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float2 t = st[min(wrow0 + i * 16 + lrow, p.m - 1)];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float4 s4 = *reinterpret_cast<const float4*>(cs + min(wcol0 + j * 16 + lq * 4, p.n - 4));
        acc[i][j][0] = t.y * (acc[i][j][0] - t.x * s4.x); acc[i][j][1] = t.y * (acc[i][j][1] - t.x * s4.y);
        acc[i][j][2] = t.y * (acc[i][j][2] - t.x * s4.z); acc[i][j][3] = t.y * (acc[i][j][3] - t.x * s4.w);
      }
    }
  }

  if (EPI == VX_EPI_STORE && nsplit > 1) {
    // split-K slice: raw fp32 partial sums to the workspace, 16 bytes (4 columns) per lane
    float* ws = (float*)p.splitk_ws + (size_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = wrow0 + i * 16 + lrow;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = wcol0 + j * 16 + lq * 4;
        if (m < p.m && n < p.n)
          *reinterpret_cast<float4*>(ws + (size_t)m * p.n + n) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
  if constexpr (EPI == VX_EPI_STORE) {
    // Optional addends are fetched in batches of NJ fragments under kernel-uniform branches (clamped addresses for
    // out-of-range rows / columns), so the loads of a row group are in flight together; only stores are predicated.
    const float* __restrict__ rowbias = p.rowbias;
    const bf16_t* __restrict__ resid = (const bf16_t*)p.residual;
    const bool do_silu = p.act == VX_ACT_SILU, do_gelu = p.act == VX_ACT_GELU;
    const bool out_is_f32 = p.out_f32 != 0;
    const float alpha = p.alpha;
    // GNS: this lane's NI x 4 columns, summed over its MI rows
    float gcs[GNS ? NI * 4 : 1], gcq[GNS ? NI * 4 : 1];
    if constexpr (GNS) {
#pragma unroll
      for (int c = 0; c < NI * 4; ++c) gcs[c] = gcq[c] = 0.f;
    }
#pragma unroll
    for (int j0 = 0; j0 < NI; j0 += NJ) {
      int ncol[NJ], nc[NJ];
      float4 bv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ncol[j] = wcol0 + (j0 + j) * 16 + lq * 4;
        nc[j] = min(ncol[j], p.n - 4);
        bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (bias != nullptr && !ABL(2)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + nc[j]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = wrow0 + i * 16 + lrow;
        const int mc = min(m, p.m - 1);
        uint2 rv[NJ];
        float4 rb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          rv[j] = make_uint2(0u, 0u);
          rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (resid != nullptr && !ABL(2)) {
          const bf16_t* rrow = resid + (size_t)mc * p.ldr;
#pragma unroll
          for (int j = 0; j < NJ; ++j) rv[j] = *reinterpret_cast<const uint2*>(rrow + nc[j]);
        }
        if (rowbias != nullptr && !ABL(2)) {
          const float* rbrow = rowbias + (size_t)(mc / p.rows_per_group) * p.rowbias_ld;
#pragma unroll
          for (int j = 0; j < NJ; ++j) rb[j] = *reinterpret_cast<const float4*>(rbrow + nc[j]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4_t a4 = acc[i][j0 + j];
          float v[4] = {a4[0] + bv[j].x + rb[j].x, a4[1] + bv[j].y + rb[j].y, a4[2] + bv[j].z + rb[j].z,
                        a4[3] + bv[j].w + rb[j].w};
          if (do_silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (do_gelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= alpha;
          v[0] += e16_lo(rv[j].x); v[1] += e16_hi(rv[j].x);
          v[2] += e16_lo(rv[j].y); v[3] += e16_hi(rv[j].y);
          if constexpr (GNS) {
            // whole tiles, bf16 output (vx_gemm_gn_slabs): statistics of the STORED values, as a read-back would see them
            const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            *reinterpret_cast<uint2*>((bf16_t*)p.out + (size_t)m * p.ldc + ncol[j]) = pk;
            const float g[4] = {e16_lo(pk.x), e16_hi(pk.x),
                                e16_lo(pk.y), e16_hi(pk.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              gcs[(j0 + j) * 4 + e] += g[e];
              gcq[(j0 + j) * 4 + e] = fmaf(g[e], g[e], gcq[(j0 + j) * 4 + e]);
            }
          } else if (ABL(1)) {
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          } else if (m < p.m && ncol[j] < p.n) {
            if (out_is_f32) {
              *reinterpret_cast<float4*>((float*)p.out + (size_t)m * p.ldc + ncol[j]) =
                  make_float4(v[0], v[1], v[2], v[3]);
            } else {
              *reinterpret_cast<uint2*>((bf16_t*)p.out + (size_t)m * p.ldc + ncol[j]) =
                  make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
          }
        }
      }
    }
    if constexpr (GNS) {
      // The 16 lanes of a DPP row hold the rows lrow of the same columns: rotate-and-add leaves the 64-row column sums in
      // every lane; lanes lrow == 0 park them in this wave's WN x 8 bytes of LDS BEHIND the pipeline stages (other waves
      // may still be reading their last K-tile), and lane g < WN / cg adds the cg columns of group g in ascending order
      // (a group never straddles a wave: WN % cg == 0 is part of vx_gemm_gn_slabs).  Same-wave LDS writes and reads are
      // ordered in the LDS queue; no barrier.
#pragma unroll
      for (int c = 0; c < NI * 4; ++c) {
        gcs[c] = row16_sum(gcs[c]);
        gcq[c] = row16_sum(gcq[c]);
      }
      float2* scr = reinterpret_cast<float2*>(smem + STAGES * STAGE_BYTES) + wave * WN;
      if (lrow == 0) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) scr[j * 16 + lq * 4 + e] = make_float2(gcs[j * 4 + e], gcq[j * 4 + e]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int cg = p.n / p.gn_groups;
      if (lane < WN / cg) {
        float a = 0.f, b = 0.f;
        for (int c = 0; c < cg; ++c) {
          const float2 t = scr[lane * cg + c];
          a += t.x;
          b += t.y;
        }
        const int frame = wrow0 / p.gn_hw, slab = (wrow0 - frame * p.gn_hw) >> 6, slabs = p.gn_hw >> 6;
        const int g = wcol0 / cg + lane;
        reinterpret_cast<float2*>(p.gn_ws)[(size_t)(frame * slabs + slab) * p.gn_groups + g] = make_float2(a, b);
      }
    }
  } else if constexpr (EPI == VX_EPI_GEGLU) {
    // Weight rows are interleaved in blocks of 8 (weights.py: geglu_interleave): columns 0-7 of a 16-column fragment
    // are the VALUES of 8 output channels, columns 8-15 their GATES, so lanes 0-31 hold values and lanes 32-63 the
    // matching gates.  v_permlane32_swap over two row blocks (i, i+1) hands lanes 0-31 value and gate of block i and
    // lanes 32-63 those of block i+1 (every lane works in the GELU); stores are 8 bytes (4 outputs) per lane.
    static_assert(MI % 2 == 0, "GEGLU pairs row blocks");
    const int nout = p.n / 2;
    const int orow0 = wrow0 + lrow + 16 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nfrag = wcol0 + j * 16;            // interleaved weight row of the fragment's first value column
      const int ocol = nfrag / 2 + 4 * (lq & 1);
      const bool col_ok = nfrag + 16 <= p.n && ocol < nout;
      float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr && col_ok) {
        const float4 h4 = *reinterpret_cast<const float4*>(bias + nfrag + 4 * (lq & 1));
        const float4 g4 = *reinterpret_cast<const float4*>(bias + nfrag + 8 + 4 * (lq & 1));
        bh[0] = h4.x; bh[1] = h4.y; bh[2] = h4.z; bh[3] = h4.w;
        bg[0] = g4.x; bg[1] = g4.y; bg[2] = g4.z; bg[3] = g4.w;
      }
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][r]), __float_as_uint(acc[i + 1][j][r]),
                                                     false, false);
          const float val = __uint_as_float(sw[0]) + bh[r], gat = __uint_as_float(sw[1]) + bg[r];
          o[r] = val * (ABL(64) ? gat : gelu_f(gat));
        }
        const int m = orow0 + 16 * i;
        if (m >= p.m || !col_ok) continue;
        if (ABL(1)) {
          asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
          continue;
        }
        *reinterpret_cast<uint2*>((bf16_t*)p.out + (size_t)m * p.ldc + ocol) =
            make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
      }
    }
  } else {  // VX_EPI_SPLIT
    // (a) row-major parts: direct 8-byte stores (a 16-column fragment never straddles parts: part_cols % 16 == 0)
    bool any_vt = false;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = wcol0 + j * 16 + lq * 4;
      const int nfrag = wcol0 + j * 16;
      if (nfrag >= p.n) continue;
      const int part = nfrag / p.part_cols;
      if (sel3(part, p.part_kind[0], p.part_kind[1], p.part_kind[2]) != VX_PART_ROWS) {
        any_vt = true;
        continue;
      }
      const int nn = n - part * p.part_cols;
      bf16_t* dst = (bf16_t*)sel3(part, p.part_out[0], p.part_out[1], p.part_out[2]);
      const int ldp = sel3(part, p.part_ld[0], p.part_ld[1], p.part_ld[2]);
      float b[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = wrow0 + i * 16 + lrow;
        if (m >= p.m) continue;
        *reinterpret_cast<uint2*>(dst + (size_t)m * ldp + nn) =
            make_uint2(pack_bf16x2(acc[i][j][0] + b[0], acc[i][j][1] + b[1]),
                       pack_bf16x2(acc[i][j][2] + b[2], acc[i][j][3] + b[3]));
      }
    }
    // (b) transposed (V^T) parts: [seq, head, dim, key] with keys contiguous.  The tile goes through LDS as
    // ct[col][row] (fp32, row-contiguous) in 64-row slabs, then 8 consecutive tokens per 16-byte store with the 8
    // token groups of a column on consecutive threads (128 contiguous bytes).
    const int tile_has_vt = __syncthreads_or(any_vt ? 1 : 0);   // also: all fragment reads of the K loop are done
    if (tile_has_vt) {
      constexpr int CT_LD = 64 + 4;   // floats per column
      float* ct = reinterpret_cast<float*>(smem);
      for (int pass = 0; pass < WARPS_M; ++pass) {
        if (wm == pass) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                ct[(wn * WN + j * 16 + lq * 4 + r) * CT_LD + i * 16 + lrow] = acc[i][j][r];
        }
        __syncthreads();
        const int mbase = m0 + pass * 64;
        for (int idx = tid; idx < 8 * BN; idx += NTHREADS) {
          const int col = idx >> 3, rg = idx & 7;
          const int n = n0 + col;
          const int mfirst = mbase + rg * 8;
          if (n >= p.n || mfirst >= p.m) continue;
          const int part = n / p.part_cols;
          if (sel3(part, p.part_kind[0], p.part_kind[1], p.part_kind[2]) != VX_PART_VT) continue;
          const int nn = n - part * p.part_cols;
          const int head = nn / p.head_dim, dd = nn - head * p.head_dim;
          const int heads = p.part_cols / p.head_dim;
          const float bv = bias != nullptr ? bias[n] : 0.f;
          const float4 lo = *reinterpret_cast<const float4*>(ct + col * CT_LD + rg * 8);
          const float4 hi = *reinterpret_cast<const float4*>(ct + col * CT_LD + rg * 8 + 4);
          float v[8] = {lo.x + bv, lo.y + bv, lo.z + bv, lo.w + bv, hi.x + bv, hi.y + bv, hi.z + bv, hi.w + bv};
          bf16_t* vt = (bf16_t*)sel3(part, p.part_out[0], p.part_out[1], p.part_out[2]);
          const int seq = mfirst / p.seq_len, tok = mfirst - seq * p.seq_len;
          if ((p.seq_len & 7) == 0 && mfirst + 8 <= p.m) {
            const size_t off = ((size_t)(seq * heads + head) * p.head_dim + dd) * p.vt_pitch + tok;
            *reinterpret_cast<uint4*>(vt + off) = pack_bf16x8(v);
          } else {
            for (int e = 0; e < 8; ++e) {
              const int m = mfirst + e;
              if (m >= p.m) break;
              const int s2 = m / p.seq_len, t = m - s2 * p.seq_len;
              vt[((size_t)(s2 * heads + head) * p.head_dim + dd) * p.vt_pitch + t] = f32_to_bf16(v[e]);
            }
          }
        }
        __syncthreads();
      }
    }
  }
}

// Second launch of a split-K GEMM: out = epilogue(sum over slices, in slice order), 4 columns per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const vx_gemm_params p) {
  const int n4 = p.n >> 2;
  const long total = (long)p.m * n4;
  const float* __restrict__ ws = (const float*)p.splitk_ws;
  const size_t slab = (size_t)p.m * p.n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    float4 a = *reinterpret_cast<const float4*>(ws + (size_t)m * p.n + n);
    for (int s = 1; s < p.splitk; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(ws + s * slab + (size_t)m * p.n + n);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias != nullptr) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.rowbias != nullptr) {
      const float4 b =
          *reinterpret_cast<const float4*>(p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.act == VX_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == VX_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
    if (p.residual != nullptr) {
      const uint2 r2 = *reinterpret_cast<const uint2*>((const bf16_t*)p.residual + (size_t)m * p.ldr + n);
      v[0] += e16_lo(r2.x); v[1] += e16_hi(r2.x);
      v[2] += e16_lo(r2.y); v[3] += e16_hi(r2.y);
    }
    if (p.out_f32) {
      *reinterpret_cast<float4*>((float*)p.out + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      *reinterpret_cast<uint2*>((bf16_t*)p.out + (size_t)m * p.ldc + n) =
          make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int EPI, bool FAST, bool F8 = false, bool LNF = false,
          bool GNS = false>
int launch_impl(const vx_gemm_params& p, hipStream_t stream) {
  constexpr int stage_bytes = STAGES * (BM + BN) * 128;
  constexpr int epi_bytes = (EPI == VX_EPI_SPLIT) ? BN * (64 + 4) * 4 : 0;   // V^T transposition slab
  constexpr int gns_bytes = GNS ? WARPS_M * BN * 8 : 0;                       // column sums of every wave, behind the stages
  constexpr int smem = (stage_bytes > epi_bytes ? stage_bytes : epi_bytes) + gns_bytes;
  constexpr int nthreads = 64 * WARPS_M * WARPS_N;
  static bool attr_set = false;
  auto kern = gemm_kernel<BM, BN, WARPS_M, WARPS_N, STAGES, EPI, FAST, F8, LNF, GNS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vx_set_error("vx_gemm: hipFuncSetAttribute(%d B LDS) failed: %s", smem, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  const int nsplit = (EPI == VX_EPI_STORE && p.splitk > 1) ? p.splitk : 1;
  long tiles = (long)ceil_div(p.m, BM) * ceil_div(p.n, BN) * nsplit;
  static char sym[112] = "";
  if (!sym[0]) {
    auto b = [](bool v) { return v ? "true" : "false"; };
    snprintf(sym, sizeof(sym), "gemm_kernel<%d, %d, %d, %d, %d, %d, %s, %s, %s, %s>", BM, BN, WARPS_M, WARPS_N, STAGES, EPI,
             b(FAST), b(F8), b(LNF), b(GNS));
  }
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(nthreads), smem, stream, p);
  int rc = vx_check_launch("vx_gemm");
  if (rc != 0 || nsplit == 1) return rc;
  long work = ((long)p.m * (p.n >> 2) + 255) / 256;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(work < 2048 ? work : 2048)), dim3(256), 0, stream, p);
  return vx_check_launch("vx_gemm(split-K reduce)");
}

// FAST-path eligibility (see gemm_kernel)
bool fast_ok(const vx_gemm_params& p) { return vx_gemm_fast_ok(p); }
}  // namespace
bool vx_gemm_fast_ok(const vx_gemm_params& p) {
  static int disabled = -1;
  if (disabled < 0) disabled = getenv("VX_GEMM_NOFAST") != nullptr;
  if (disabled) return false;
  const int cin = p.c1 + p.c2;
  const unsigned long long rows_in = (unsigned long long)p.nb * p.h_in * p.w_in;
  return p.pad == 0 && p.upsample == 0 && (cin % 64) == 0 && (p.c1 % 64) == 0 && (p.k % 64) == 0 &&
         rows_in * (unsigned long long)p.lda1 * 2ull < (1ull << 32) &&
         (p.c2 == 0 || rows_in * (unsigned long long)p.lda2 * 2ull < (1ull << 32)) &&
         (unsigned long long)p.n * p.k * 2ull < (1ull << 32) &&
         (p.h_out - 1) * p.stride + p.kh <= p.h_in && (p.w_out - 1) * p.stride + p.kw <= p.w_in;
}
namespace {

template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int EPI>
int launch(const vx_gemm_params& p, hipStream_t stream) {
  if (p.ln_stats != nullptr) {
    if (!fast_ok(p)) {
      vx_set_error("vx_gemm: a folded LayerNorm needs a plain linear (k %% 64 == 0, no padding / upsampling)");
      return VX_ERR_UNSUPPORTED;
    }
    return launch_impl<BM, BN, WARPS_M, WARPS_N, STAGES, EPI, true, false, true>(p, stream);
  }
  if (fast_ok(p)) return launch_impl<BM, BN, WARPS_M, WARPS_N, STAGES, EPI, true>(p, stream);
  return launch_impl<BM, BN, WARPS_M, WARPS_N, STAGES, EPI, false>(p, stream);
}

// STORE launch that also writes GroupNorm partial sums (p.gn_ws): instantiated for the tiles the 16x16 / 8x8 levels use
template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES>
int launch_gns(const vx_gemm_params& p, hipStream_t stream) {
  if (fast_ok(p)) return launch_impl<BM, BN, WARPS_M, WARPS_N, STAGES, VX_EPI_STORE, true, false, false, true>(p, stream);
  return launch_impl<BM, BN, WARPS_M, WARPS_N, STAGES, VX_EPI_STORE, false, false, false, true>(p, stream);
}

// column-tile width with the least padding (ties -> 160: fewer, fatter tiles)
bool prefer160(int n) {
  int w160 = ceil_div(n, 160) * 160 - n, w128 = ceil_div(n, 128) * 128 - n;
  return w160 * 128 <= w128 * 160;
}

// Tile configuration.  BIG = 256x320 tile, 8 waves, one block per CU: A is streamed once for N = 320 and a
// K-tile's DMA (72 KiB) is covered by 2x the MFMA work of the 128-row tiles -> used when N is a multiple of 320 and
// the launch still has >= 1 tile per CU.  VX_GEMM_TILE=big|small|small3 overrides (benchmarking only).
enum { CFG_AUTO = 0, CFG_BIG = 1, CFG_SMALL = 2 };
int forced_cfg() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VX_GEMM_TILE");
    v = CFG_AUTO;
    if (e && !strcmp(e, "big")) v = CFG_BIG;
    if (e && !strcmp(e, "small")) v = CFG_SMALL;
  }
  return v;
}
bool use_big(const vx_gemm_params& p) {
  if ((p.n % 320) != 0 || p.splitk > 1) return false;   // split-K wants many small tiles
  int f = forced_cfg();
  if (f == CFG_BIG) return true;
  if (f == CFG_SMALL) return false;
  long tiles = (long)ceil_div(p.m, 256) * (p.n / 320);
  return tiles >= 256;
}

// 64 x 160 tile, 2 waves, 3 stages: launches whose 128-row tiling would leave CUs without a block (the 8x8 level of a
// 16-frame window: M = 2048 -> 128 tiles of 128 x 160 for N = 1280).  Which tile a launch gets never changes its
// results: every tile shape accumulates an output element over the K-tiles in the same order with the same MFMA.
// VX_GEMM_SMALL64=0 disables (A/B measurements).
bool use_small64(const vx_gemm_params& p) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VX_GEMM_SMALL64");
    on = !(e && !strcmp(e, "0"));
  }
  if (!on || !prefer160(p.n) || p.out_f32) return false;
  const long tiles128 = (long)ceil_div(p.m, 128) * ceil_div(p.n, 160) * (p.splitk > 1 ? p.splitk : 1);
  static long lim = -1;   // VX_GEMM_SMALL64_BELOW (A/B knob): the 64-row tile is used while the 128-row tiling has fewer blocks
  if (lim < 0) {
    const char* e = getenv("VX_GEMM_SMALL64_BELOW");
    lim = e ? atol(e) : 256;
  }
  return tiles128 < lim && fast_ok(p);
}

// which classic tile a bf16-operand STORE launch gets (the persistent ring kernel is asked first by the callers)
enum { T_256x32 = 0, T_BIG, T_SMALL64, T_128x160, T_128x128, T_128x320, T_256x256 };
// 256 x 256 tile, 8 waves as 4 x 2 (wave 64 x 128), one block per CU: the VAE decoder's 256- and 512-channel convolutions
// (M = 65536 ... 1048576 rows): 7.8 KB of operands per MFLOP through the CU's L1 instead of the 128 x 128 tile's 15.6.
// VX_GEMM_T256X256=0 keeps them on the 128 x 128 tile (A/B knob).
bool use_256x256(const vx_gemm_params& p) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VX_GEMM_T256X256");
    on = !(e && !strcmp(e, "0"));
  }
  return on && (p.n % 256) == 0 && p.n <= 1024 && p.splitk <= 1 && !p.out_f32 && (long)ceil_div(p.m, 256) * (p.n / 256) >= 256;
}
// VX_GEMM_T128X320=1 (experiment, round 4): the 16x16-level launches (n % 320 == 0, too few 256-row tiles for the big
// kernels) on a 128 x 320 tile, 8 waves as 2 x 4, one block per CU: 10.9 KB of operands per MFLOP through the CU's L1
// instead of the 128 x 160 tile's 14.1
bool use_128x320(const vx_gemm_params& p) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VX_GEMM_T128X320");
    on = e && !strcmp(e, "1");
  }
  return on && (p.n % 320) == 0 && p.splitk <= 1 && !p.out_f32;
}
// gn: the launch is asked for GroupNorm partial sums (vx_gemm_params.gn_ws).  Only the 64-row-per-wave 64 x 160 and
// 128 x 160 tiles produce them, and WHETHER a launch produces them must not depend on how many frames share it (the sums
// then come from a differently grouped statistics pass: other bits) - so a request for them takes the launch off the
// 256 x 320 tile, whose use depends on the row count.
int store_tile(const vx_gemm_params& p, bool gn = false) {
  if (p.n <= 32) return T_256x32;
  if (gn && (p.n % 160) == 0) return use_small64(p) ? T_SMALL64 : T_128x160;
  if (use_big(p)) return T_BIG;
  if (use_small64(p)) return T_SMALL64;
  if (use_128x320(p)) return T_128x320;
  if (prefer160(p.n)) return T_128x160;
  if (use_256x256(p)) return T_256x256;
  return T_128x128;
}

}  // namespace

// GroupNorm partial sums from the STORE epilogue (vx_gemm_params.gn_ws): slabs per frame the launch of p writes, 0 = it
// cannot (the caller keeps the separate statistics pass).  Mirrors vx_gemm_dispatch's kernel choice.
extern "C" int vx_gemm_gn_slabs(const vx_gemm_params* pp) {
  const vx_gemm_params& p = *pp;
  if (p.epi != VX_EPI_STORE || p.a_fp8 || p.out_f32 || p.ln_stats != nullptr || p.row_stats_out != nullptr ||
      (p.splitk > 1 && p.ring_hint != 2) || p.w_group_rows != 0)
    return 0;
  if (p.gn_groups <= 0 || p.gn_hw <= 0 || p.n <= 0 || (p.n % p.gn_groups) != 0 || (p.m % p.gn_hw) != 0) return 0;
  if (vx_gemm_ring_eligible(p)) return vx_gemm_ring_gn_slabs(p);
  if (p.splitk > 1) return 0;
  const int cg = p.n / p.gn_groups;
  const int tile = store_tile(p, true);
  if (tile != T_SMALL64 && tile != T_128x160) return 0;
  static int st128 = -1;
  if (st128 < 0) {
    const char* e = getenv("VX_GEMM_STAGES128");
    st128 = e ? atoi(e) : 2;
  }
  if (tile == T_128x160 && st128 != 2) return 0;   // (the A/B depth knob has no partial-sum instantiation)
  const int bm = tile == T_SMALL64 ? 64 : 128;
  if ((p.m % bm) != 0 || (p.n % 160) != 0 || (p.gn_hw % 64) != 0 || (80 % cg) != 0) return 0;   // whole tiles, wave = 64 x 80
  return p.gn_hw / 64;
}

thread_local const char* g_vx_last_kernel = "";
extern "C" const char* vx_gemm_last_kernel(void) { return g_vx_last_kernel; }
extern "C" const char* vx_last_kernel(void) { return g_vx_last_kernel; }

// whether the launch described by *pp could run as the persistent kernel's cooperative two-way K split (ring_hint = 2,
// splitk = 2, splitk_ws = zeroed vx_gemm_splitk_ws_bytes(m, n, 2) bytes): pp's own ring_hint / splitk / splitk_ws are ignored
extern "C" int vx_gemm_ring_coop_ok(const vx_gemm_params* pp) {
  vx_gemm_params q = *pp;
  q.ring_hint = 2;
  q.splitk = 2;
  if (q.splitk_ws == nullptr) q.splitk_ws = (void*)16;   // (only tested for null)
  return vx_gemm_ring_eligible(q) ? 1 : 0;
}

extern "C" int64_t vx_gemm_splitk_ws_bytes(int m, int n, int splitk) {
  return splitk > 1 ? (int64_t)splitk * m * n * (int64_t)sizeof(float) : 0;
}

extern "C" const char* vx_gemm_config_name(const vx_gemm_params* pp) {
  const vx_gemm_params& p = *pp;
  const bool fast = fast_ok(p);
  const char* epi = p.epi == VX_EPI_STORE ? "STORE" : (p.epi == VX_EPI_GEGLU ? "GEGLU" : "SPLIT");
  const char* tile;
  if (p.a_fp8) {
    static thread_local char b8[96];
    if (p.epi == VX_EPI_STORE && vx_gemm_ring_eligible(p)) return "gemm_ring_kernel<256x320x128,8w,STORE,fast,fp8>";
    const bool big8 = (p.n % 320) == 0 && (long)ceil_div(p.m, 256) * (p.n / 320) >= 256;
    snprintf(b8, sizeof(b8), "gemm_kernel<%s,%s,fast,fp8>", big8 ? "256x320x128,8w" : "128x160x128,4w", epi);
    return b8;
  }
  if (vx_gemm_ring_eligible(p))
    return p.epi == VX_EPI_GEGLU ? "gemm_ring_kernel<256x320x64,8w,GEGLU,fast>"
                                 : (p.ring_hint == 2 ? "gemm_ring_kernel<256x320x64,8w,STORE,fast,coop2>"
                                                     : "gemm_ring_kernel<256x320x64,8w,STORE,fast>");
  if (p.epi == VX_EPI_STORE && p.n <= 32) tile = "256x32x64,4w";
  else if (use_big(p)) tile = "256x320x64,8w";
  else if (p.epi == VX_EPI_STORE && use_small64(p)) tile = "64x160x64,2w";
  else if (p.epi == VX_EPI_STORE && use_128x320(p)) tile = "128x320x64,8w";
  else if (p.epi == VX_EPI_STORE && !prefer160(p.n) && use_256x256(p)) tile = "256x256x64,8w";
  else if (p.epi != VX_EPI_GEGLU && prefer160(p.n)) tile = "128x160x64,4w";
  else tile = "128x128x64,4w";
  static thread_local char buf[96];
  if (p.splitk > 1)
    snprintf(buf, sizeof(buf), "gemm_kernel<%s,%s,%s,splitk%d>", tile, epi, fast ? "fast" : "gather", p.splitk);
  else
    snprintf(buf, sizeof(buf), "gemm_kernel<%s,%s,%s>", tile, epi, fast ? "fast" : "gather");
  return buf;
}

static int vx_gemm_dispatch(const vx_gemm_params& p, hipStream_t stream);

extern "C" int vx_gemm(const vx_gemm_params* pp, void* stream_) {
  const vx_gemm_params& p = *pp;
  hipStream_t stream = (hipStream_t)stream_;
  if (p.row_stats_out != nullptr)
    VX_REQUIRE(p.epi == VX_EPI_STORE && !p.out_f32 && p.row_stats_eps > 0.f && p.out != nullptr,
               "vx_gemm: row_stats_out needs the STORE epilogue into bf16 and row_stats_eps > 0");
  VX_REQUIRE(p.row_stats_parts == 0 || p.row_stats_parts == 1 || (p.row_stats_parts == 2 && p.n == 640),
             "vx_gemm: row_stats_parts=%d (0 / 1, or 2 with n == 640; n=%d)", p.row_stats_parts, p.n);
  if (p.ln_stats != nullptr && p.ln_stats_parts == 2) {
    if (p.k != 640 || !(p.ln_eps > 0.f) || !vx_gemm_ring_eligible(p)) {
      vx_set_error("vx_gemm: ln_stats_parts = 2 needs k == 640, ln_eps > 0 and a launch on the persistent kernel (k=%d m=%d "
                   "n=%d): convert with vx_row_stats_finalize", p.k, p.m, p.n);
      return VX_ERR_UNSUPPORTED;
    }
  } else {
    VX_REQUIRE(p.ln_stats_parts == 0 || p.ln_stats_parts == 1 || p.ln_stats == nullptr,
               "vx_gemm: ln_stats_parts=%d (0 / 1 / 2)", p.ln_stats_parts);
  }
  if (p.w_group_rows != 0)
    VX_REQUIRE(p.w_group_rows > 0 && (p.m % p.w_group_rows) == 0 && p.epi == VX_EPI_STORE && !p.a_fp8 && p.splitk <= 1,
               "vx_gemm: w_group_rows=%d must divide m=%d (STORE epilogue, no fp8 / split-K)", p.w_group_rows, p.m);
  if (p.gn_ws != nullptr && vx_gemm_gn_slabs(pp) <= 0) {
    vx_set_error("vx_gemm: this launch cannot produce GroupNorm partial sums (gn_ws set; vx_gemm_gn_slabs() == 0: m=%d "
                 "n=%d groups=%d hw=%d epi=%d splitk=%d)", p.m, p.n, p.gn_groups, p.gn_hw, p.epi, p.splitk);
    return VX_ERR_UNSUPPORTED;
  }
  const int rc = vx_gemm_dispatch(p, stream);
  if (rc != VX_OK || p.row_stats_out == nullptr) return rc;
  if (vx_gemm_ring_eligible(p) && vx_gemm_ring_writes_row_stats(p)) return rc;   // the epilogue wrote them
  if (p.row_stats_parts == 2) return vx_row_stats_parts(p.out, p.ldc, p.m, p.n, p.row_stats_out, stream_);
  return vx_row_stats(p.out, p.ldc, p.m, p.n, p.row_stats_eps, p.row_stats_out, stream_);
}

static int vx_gemm_dispatch(const vx_gemm_params& p, hipStream_t stream) {
  VX_REQUIRE(p.a != nullptr && p.w != nullptr, "vx_gemm: null operand");
  VX_REQUIRE(p.m > 0 && p.n > 0 && p.k > 0, "vx_gemm: empty problem m=%d n=%d k=%d", p.m, p.n, p.k);
  VX_REQUIRE((p.c1 % 8) == 0 && (p.c2 % 8) == 0 && p.c1 > 0, "vx_gemm: channels must be multiples of 8 (c1=%d c2=%d)",
             p.c1, p.c2);
  VX_REQUIRE((p.c2 == 0) == (p.a2 == nullptr), "vx_gemm: a2/c2 mismatch");
  VX_REQUIRE(p.k == p.kh * p.kw * (p.c1 + p.c2), "vx_gemm: k=%d != kh*kw*(c1+c2)=%d", p.k,
             p.kh * p.kw * (p.c1 + p.c2));
  VX_REQUIRE(p.m == p.nb * p.h_out * p.w_out, "vx_gemm: m=%d != nb*h_out*w_out", p.m);
  VX_REQUIRE((p.n % 8) == 0, "vx_gemm: n=%d must be a multiple of 8", p.n);
  VX_REQUIRE((p.lda1 % 8) == 0 && (p.c2 == 0 || (p.lda2 % 8) == 0), "vx_gemm: lda must be a multiple of 8");
  VX_REQUIRE(p.upsample == 0 || p.upsample == 1, "vx_gemm: upsample must be 0/1");
  VX_REQUIRE(p.stride >= 1 && p.kh >= 1 && p.kw >= 1, "vx_gemm: bad conv geometry");
  VX_REQUIRE(p.ln_stats == nullptr || (p.ln_colsum != nullptr && p.splitk <= 1 && !p.a_fp8),
             "vx_gemm: a folded LayerNorm needs ln_colsum and excludes split-K / fp8 operands");
  VX_REQUIRE(p.splitk <= 1 || (p.epi == VX_EPI_STORE && p.splitk_ws != nullptr && p.splitk <= 16 &&
                               p.splitk <= (p.k + BK - 1) / BK),
             "vx_gemm: split-K needs the STORE epilogue, a workspace and splitk <= min(16, K/64)");
  if (p.a_fp8) {
    // fp8 projections: plain linears over zero-padded K (see vx_gemm_params.a_fp8)
    VX_REQUIRE(p.a_scale != nullptr && p.w_scale != nullptr, "vx_gemm(fp8): null scale table");
    VX_REQUIRE(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && p.upsample == 0 && p.a2 == nullptr &&
                   p.splitk <= 1 && p.h_out == p.h_in && p.w_out == p.w_in,
               "vx_gemm(fp8): plain linears only");
    VX_REQUIRE((p.k % 128) == 0 && (p.lda1 % 16) == 0 && p.c1 == p.k && (p.n % 4) == 0,
               "vx_gemm(fp8): k=%d must be a multiple of 128 (zero-padded), lda1=%d of 16", p.k, p.lda1);
    VX_REQUIRE((unsigned long long)p.m * p.lda1 < (1ull << 32) && (unsigned long long)p.n * p.k < (1ull << 32),
               "vx_gemm(fp8): operand spans 4 GiB");
    VX_REQUIRE(p.epi == VX_EPI_STORE || p.epi == VX_EPI_SPLIT, "vx_gemm(fp8): STORE / SPLIT epilogues only");
    const bool big = (p.n % 320) == 0 && (long)ceil_div(p.m, 256) * (p.n / 320) >= 256;
    if (p.epi == VX_EPI_STORE) {
      VX_REQUIRE(p.out != nullptr && (p.ldc % 8) == 0, "vx_gemm: STORE needs out and ldc%%8==0");
      VX_REQUIRE(p.residual == nullptr || (p.ldr % 8) == 0, "vx_gemm: ldr%%8");
      VX_REQUIRE(p.rowbias == nullptr || p.rows_per_group > 0, "vx_gemm: rows_per_group");
      if (vx_gemm_ring_eligible(p)) return vx_gemm_ring_launch(p, stream);
      if (big) return launch_impl<256, 320, 4, 2, 2, VX_EPI_STORE, true, true>(p, stream);
      return launch_impl<128, 160, 2, 2, 2, VX_EPI_STORE, true, true>(p, stream);
    }
  }
  if (p.epi == VX_EPI_STORE) {
    VX_REQUIRE(p.out != nullptr && (p.ldc % 8) == 0, "vx_gemm: STORE needs out and ldc%%8==0");
    VX_REQUIRE(p.residual == nullptr || (p.ldr % 8) == 0, "vx_gemm: ldr%%8");
    VX_REQUIRE(p.rowbias == nullptr || p.rows_per_group > 0, "vx_gemm: rows_per_group");
    if (p.ring_hint == 2 && (p.coop_epoch < 1 || p.coop_epoch >= (1 << 27))) {
      vx_set_error("vx_gemm: ring_hint = 2 needs 1 <= coop_epoch < 2^27 (the workspace's launch counter), got %d", p.coop_epoch);
      return VX_ERR_INVALID;
    }
    if (vx_gemm_ring_eligible(p)) return vx_gemm_ring_launch(p, stream);
    if (p.ring_hint == 2) {
      vx_set_error("vx_gemm: ring_hint = 2 (cooperative two-way K split on the persistent kernel) needs splitk == 2, a zeroed "
                   "workspace, m %% 256 == 0, n %% 320 == 0, an even number of 64-channel chunks and the plain STORE epilogue "
                   "(m=%d n=%d k=%d splitk=%d): ask vx_gemm_ring_coop_ok() first", p.m, p.n, p.k, p.splitk);
      return VX_ERR_UNSUPPORTED;
    }
    if (p.w_group_rows != 0) {
      vx_set_error("vx_gemm: per-row-group weights (w_group_rows=%d) need a launch the persistent 256 x 320 kernel "
                   "accepts (m %% 256, n %% 320, w_group_rows %% 256, plain addressing)", p.w_group_rows);
      return VX_ERR_UNSUPPORTED;
    }
    const int tile = store_tile(p, p.gn_ws != nullptr);
    if (p.gn_ws != nullptr) {
      // (vx_gemm checked vx_gemm_gn_slabs(p) > 0: only these two tiles produce the partial sums)
      if (tile == T_SMALL64) return launch_gns<64, 160, 1, 2, 3>(p, stream);
      return launch_gns<128, 160, 2, 2, 2>(p, stream);
    }
    if (tile == T_256x32) return launch<256, 32, 4, 1, 2, VX_EPI_STORE>(p, stream);
    if (tile == T_128x320) return launch<128, 320, 2, 4, 2, VX_EPI_STORE>(p, stream);
    if (tile == T_256x256) return launch<256, 256, 4, 2, 2, VX_EPI_STORE>(p, stream);
    if (tile == T_BIG) return launch<256, 320, 4, 2, 2, VX_EPI_STORE>(p, stream);
    if (tile == T_SMALL64) return launch<64, 160, 1, 2, 3, VX_EPI_STORE>(p, stream);
    if (tile == T_128x160) {
      // pipeline depth of the 128 x 160 tile (VX_GEMM_STAGES128 = 2 | 3 | 4; A/B knob): the 16x16-level launches are
      // DMA-latency bound at depth 2 (8192 x 1280 x 1280: 2 us per K-tile against 0.3 us of MFMA work)
      static int st = -1;
      if (st < 0) {
        const char* e = getenv("VX_GEMM_STAGES128");
        st = e ? atoi(e) : 2;
      }
      if (st == 3) return launch<128, 160, 2, 2, 3, VX_EPI_STORE>(p, stream);
      if (st == 4) return launch<128, 160, 2, 2, 4, VX_EPI_STORE>(p, stream);
      return launch<128, 160, 2, 2, 2, VX_EPI_STORE>(p, stream);
    }
    return launch<128, 128, 2, 2, 2, VX_EPI_STORE>(p, stream);
  } else if (p.epi == VX_EPI_GEGLU) {
    VX_REQUIRE(p.out != nullptr && (p.n % 32) == 0 && (p.ldc % 8) == 0,
               "vx_gemm: GEGLU needs n%%32==0 (value/gate rows interleaved in blocks of 8)");
    if (vx_gemm_ring_eligible(p)) return vx_gemm_ring_launch(p, stream);
    if (use_big(p)) return launch<256, 320, 4, 2, 2, VX_EPI_GEGLU>(p, stream);
    return launch<128, 128, 2, 2, 2, VX_EPI_GEGLU>(p, stream);
  } else if (p.epi == VX_EPI_SPLIT) {
    VX_REQUIRE(p.n_parts >= 1 && p.n_parts <= 3 && p.part_cols > 0 && p.n == p.n_parts * p.part_cols,
               "vx_gemm: SPLIT n=%d != n_parts*part_cols", p.n);
    VX_REQUIRE((p.part_cols % 16) == 0, "vx_gemm: part_cols%%16 (a 16-column fragment must not straddle parts)");
    for (int i = 0; i < p.n_parts; ++i) {
      VX_REQUIRE(p.part_out[i] != nullptr, "vx_gemm: SPLIT part %d has no destination", i);
      if (p.part_kind[i] == VX_PART_VT)
        VX_REQUIRE(p.seq_len > 0 && p.head_dim > 0 && (p.part_cols % p.head_dim) == 0 && (p.vt_pitch % 8) == 0 &&
                       p.vt_pitch >= p.seq_len && (p.m % p.seq_len) == 0,
                   "vx_gemm: bad V^T geometry seq_len=%d head_dim=%d pitch=%d", p.seq_len, p.head_dim, p.vt_pitch);
      else
        VX_REQUIRE((p.part_ld[i] % 8) == 0, "vx_gemm: part_ld%%8");
    }
    if (p.a_fp8) {
      if ((p.n % 320) == 0 && (long)ceil_div(p.m, 256) * (p.n / 320) >= 256)
        return launch_impl<256, 320, 4, 2, 2, VX_EPI_SPLIT, true, true>(p, stream);
      return launch_impl<128, 160, 2, 2, 2, VX_EPI_SPLIT, true, true>(p, stream);
    }
    if (use_big(p)) return launch<256, 320, 4, 2, 2, VX_EPI_SPLIT>(p, stream);
    if (prefer160(p.n)) return launch<128, 160, 2, 2, 2, VX_EPI_SPLIT>(p, stream);
    return launch<128, 128, 2, 2, 2, VX_EPI_SPLIT>(p, stream);
  }
  vx_set_error("vx_gemm: unknown epilogue %d", p.epi);
  return VX_ERR_UNSUPPORTED;
}
