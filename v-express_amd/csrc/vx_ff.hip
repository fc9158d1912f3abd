// Fused GEGLU feed-forward for the 64x64 level (C = 320) on gfx950:   out = x + (GEGLU(LN(x) W1^T + b1)) W2^T + b2
// in ONE launch, without the [rows, 4C] intermediate ever leaving the CU (diffusers FeedForward(activation_fn="geglu")
// reached from modules/mutual_self_attention.py:247 and modules/motion_module.py:256; today two vx_gemm launches with a
// 335 MB round trip through HBM at the 64x64 level).  Structure = flash attention with "keys" = hidden channels:
//   P = x W1c^T (K = 320)  ->  h = value * gelu(gate)  ->  Y += h W2c^T (K = 32)      per chunk c of 32 hidden channels
//
//   * one 256-thread workgroup (4 waves, ONE per SIMD, up to 512 VGPRs each) per CU walks a list of 128-row tiles;
//     waves as 2 (M) x 2 (N): a wave owns 64 rows;
//   * the wave's x rows live in REGISTERS for the whole tile as MFMA operand fragments (4 row blocks x 10 k-steps x
//     16 B = 160 VGPRs) - with C = 320 that is what makes a 128-row tile possible at all: an x tile in LDS (80 KB) would
//     leave no room for a weight ring, and x fragments re-read from LDS per chunk would saturate the LDS port;
//   * Y (128 x 320 fp32) = 160 VGPRs per lane, P (128 x 64) = 32;
//   * the weights are pre-tiled at load time (vx_ff_pack_weights) into MFMA-fragment-major order, so a chunk is ONE
//     contiguous 40 KB (W1) / 20 KB (W2) block: the LDS-DMA copies are linear, and every fragment read is a linear 1 KB
//     ds_read_b128 (conflict-free without a swizzle);
//   * ONE barrier per chunk: stage 2 runs one chunk behind stage 1, its 40 MFMAs interleaved with the GEGLU arithmetic of
//     the current chunk (the wave is alone on its SIMD: nothing else covers the VALU work); two-buffer rings for both weight
//     streams, refilled one chunk ahead, one copy per k-step (the weights are L2-resident: every CU streams the same
//     2.4 MB); the stream wraps around (every tile reads the same weights), so it runs across tile boundaries;
//   * h goes from the GEGLU registers to the second MFMA through 8 KB of LDS written in A-fragment-major order.
// The LayerNorm in front of the FF is folded in (vx_gemm_params.ln_stats convention: w1 = gamma (.) W1, colsum, b1').
#include "vx_common.h"
#include "vx_gemm_common.h"
#include "../../include/vexpress_hip.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int FF_C = 320, FF_H = 1280, FF_BM = 128, FF_HC = 32;
constexpr int FF_CHUNKS = FF_H / FF_HC;            // 40
constexpr int FF_KS = FF_C / 32;                   // 10 k-steps of 32 in stage 1
constexpr int W1_CHUNK = 2 * FF_HC * FF_C * 2;     // 40960 B: 64 interleaved rows x 320 k
constexpr int W2_CHUNK = FF_C * FF_HC * 2;         // 20480 B: 320 rows x 32 k
constexpr int W1_OFF = 0, W2_OFF = 2 * W1_CHUNK, H_OFF = W2_OFF + 2 * W2_CHUNK;     // 0, 81920, 122880
constexpr int H_BUF = FF_BM * FF_HC * 2;           // 8192 B
constexpr int TAB_OFF = H_OFF + 2 * H_BUF;         // 139264: bias1 [2560], colsum [2560], bias2 [320] (fp32)
constexpr int FF_LDS = TAB_OFF + (2 * FF_H + 2 * FF_H + FF_C) * 4;   // 161024 <= 163840

// Compile-time ablation switches (tools/build_ff_variants.sh: one library per mask; the product library is built without):
//   1 no weight copies inside the chunk loop   2 GEGLU without the GELU (value * gate)   4 no LayerNorm fold / bias
//   8 no stage-2 MFMAs   16 no stage-1 MFMAs   32 no h round trip through LDS (no write, reads stale)
//   64 no LDS fragment reads at all (weights / h fragments are whatever the registers hold)   128 no barrier / copy wait
#ifdef VX_FF_ABLATE
#define FABL(bit) (((VX_FF_ABLATE) & (bit)) != 0)
#else
#define FABL(bit) false
#endif
// experiment knobs (tools/build_ff_variants.sh): VX_FF_PF = k-steps the W1 fragment reads run ahead of their MFMAs
// (product 2), VX_FF_NOSB = no scheduling barrier between the k-steps of segment (a)
#ifndef VX_FF_PF
#define VX_FF_PF 2
#endif
// VX_FF_S2SPLIT = 1: the eight-wave form runs stage 2 as 2 (M) x 4 (N) (wave = 64 rows x 80 columns: 9 fragment reads per
// chunk instead of 12 - the kernel is bound by its LDS fragment reads, profiles/r04e_*);  VX_FF_XPF = 1: the next tile's x
// fragments are loaded before the tail of the current tile (stage 2 of the last chunk + epilogue) instead of after it
#ifndef VX_FF_S2SPLIT
#define VX_FF_S2SPLIT 1
#endif
#ifndef VX_FF_XPF
#define VX_FF_XPF 1
#endif

__device__ __forceinline__ uint4 ff_frag(const char* p) {
  if (FABL(64)) return make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  return *reinterpret_cast<const uint4*>(p);
}

template <int N>
__device__ __forceinline__ void ff_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ff_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// MI = 16-row blocks per wave: 4 -> four waves (one per SIMD, up to 512 VGPRs: everything of a chunk has to be interleaved
// by hand inside ONE instruction stream), 2 -> eight waves as 4 (M) x 2 (N), two per SIMD (<= 256 VGPRs: x 80 + Y 80 +
// P 16), where the hardware interleaves one wave's MFMAs with its partner's GEGLU arithmetic, LDS waits and copy issue.
template <int MI>
__global__ __launch_bounds__(64 * (16 / MI), MI == 4 ? 1 : 2) void ff_fused_kernel(const vx_ff_params p) {
  constexpr int NW = 16 / MI, NT = 64 * NW;             // waves / threads per workgroup
  constexpr int W1_BLOCKS = W1_CHUNK / 1024, W2_BLOCKS = W2_CHUNK / 1024;            // 40, 20 one-KiB copies per chunk
  constexpr int W1_PER_WAVE = (W1_BLOCKS + NW - 1) / NW, W2_PER_WAVE = (W2_BLOCKS + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;              // wave = rows 16 MI wm .., columns: half wn
  const int lrow = lane & 15, lq = lane >> 4;
  const uint32_t lds0 = lds_addr_of(smem);
  const uint32_t lane16 = (uint32_t)lane * 16u;

  // ---- tables: bias1 | colsum | bias2 (fp32) once per workgroup
  {
    float* tabw = reinterpret_cast<float*>(smem + TAB_OFF);
    for (int i = tid; i < 2 * FF_H; i += NT) {
      tabw[i] = p.bias1 != nullptr ? p.bias1[i] : 0.f;
      tabw[2 * FF_H + i] = p.ln_colsum[i];
    }
    for (int i = tid; i < FF_C; i += NT) tabw[4 * FF_H + i] = p.bias2 != nullptr ? p.bias2[i] : 0.f;
  }
  __syncthreads();

  // tile walk: XCD x (= block id % 8) owns the contiguous x-th eighth of the 128-row tiles, like the GEMM and attention
  // kernels around this one (vx_gemm_ring.hip, VX_XCD_ROWS): the rows this kernel reads were written on the same XCD
  const int n_tiles = p.m / FF_BM;
  const int G = gridDim.x;
#if VX_XCD_ROWS
  const int nxcd = G < 8 ? G : 8;
  const int xcd = (int)blockIdx.x % nxcd, xidx = (int)blockIdx.x / nxcd;
  const int tstride = (G - xcd + nxcd - 1) / nxcd;
  const int t0 = (int)((long)n_tiles * xcd / nxcd), t1 = (int)((long)n_tiles * (xcd + 1) / nxcd);
  const int tfirst = t0 + xidx;
  const int my_tiles = tfirst < t1 ? (t1 - tfirst + tstride - 1) / tstride : 0;
  if (my_tiles <= 0) return;
#else
  const int tstride = G, tfirst = (int)blockIdx.x;
  const int my_tiles = (n_tiles - tfirst + G - 1) / G;
#endif
  const char* __restrict__ w1t = (const char*)p.w1t;
  const char* __restrict__ w2t = (const char*)p.w2t;

  // weight stream: global chunk counter g (chunk = g % 40, ring buffer = g & 1); this wave copies blocks wave, wave + NW, ...
  // one copy per call, spread over the k-steps of stage 1 (a burst of back-to-back copies stalls the issuing wave)
  const char* w1src = nullptr;
  const char* w2src = nullptr;
  uint32_t w1dst = 0, w2dst = 0;
  auto stream_setup = [&](int g) {      // copies issued during iteration g: W1(g + 1), W2(g)
    w1src = w1t + (size_t)((g + 1) % FF_CHUNKS) * W1_CHUNK + wave * 1024;
    w1dst = lds0 + W1_OFF + ((g + 1) & 1) * W1_CHUNK + wave * 1024;
    w2src = w2t + (size_t)(g % FF_CHUNKS) * W2_CHUNK + wave * 1024;
    w2dst = lds0 + W2_OFF + (g & 1) * W2_CHUNK + wave * 1024;
  };
  bool in_loop = false;
  auto issue_w1 = [&](int q) {
    if (FABL(1) && in_loop) return;
    if (wave + NW * q < W1_BLOCKS) glds16_s(w1src + q * (NW * 1024), lane16, w1dst + q * (NW * 1024));
  };
  auto issue_w2 = [&](int q) {
    if (FABL(1) && in_loop) return;
    if (wave + NW * q < W2_BLOCKS) glds16_s(w2src + q * (NW * 1024), lane16, w2dst + q * (NW * 1024));
  };
  // prologue: W1(0)
  {
    w1src = w1t + wave * 1024;
    w1dst = lds0 + W1_OFF + wave * 1024;
#pragma unroll
    for (int q = 0; q < W1_PER_WAVE; ++q) issue_w1(q);
  }

  const bf16_t* __restrict__ x = (const bf16_t*)p.x;
  const float2* __restrict__ st = reinterpret_cast<const float2*>(p.ln_stats);
  const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
  int g = 0;
  in_loop = true;
  // stage-2 / epilogue ownership: like stage 1 (rows 16 MI wm, columns 160 wn), or - eight waves - 2 (M) x 4 (N)
  constexpr bool S24 = (MI == 2) && (VX_FF_S2SPLIT != 0);
  constexpr int SMI = S24 ? 4 : MI, SNJ = S24 ? 5 : 10;
  const int s2_row0 = S24 ? 64 * (wave >> 2) : 16 * MI * wm;
  const int s2_col0 = S24 ? 80 * (wave & 3) : 160 * wn;
  // ---- x fragments of the wave's rows (second MFMA operand: lane = row lrow, k group lq) and their LayerNorm scalars
  uint4 xa[MI][FF_KS];
  float rs[MI], rm[MI];
  auto load_x = [&](int ti) {
    const int r0 = (tfirst + ti * tstride) * FF_BM + 16 * MI * wm;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const bf16_t* row = x + (size_t)(r0 + 16 * i + lrow) * p.ldx + 8 * lq;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) xa[i][ks] = *reinterpret_cast<const uint4*>(row + 32 * ks);
      const float2 t = st[r0 + 16 * i + lrow];
      rs[i] = t.y;
      rm[i] = -t.x * t.y;
    }
  };
  constexpr bool XPF = (VX_FF_XPF != 0) && MI == 2;   // (the four-wave form has no registers to spare)
  if (XPF) load_x(0);
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile0 = (tfirst + ti * tstride) * FF_BM;    // first row of the tile
    if (!XPF) load_x(ti);
    f32x4_t Y[SMI][SNJ];
#pragma unroll
    for (int i = 0; i < SMI; ++i)
#pragma unroll
      for (int j = 0; j < SNJ; ++j) Y[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // stage 2 of chunk (g - 1): Y += h W2c^T, as five groups of SMI * SNJ / 5 MFMAs that the caller interleaves with the GEGLU
    // arithmetic of chunk g
    uint4 ha[SMI];
    const char* w2b = nullptr;
    auto stage2_begin = [&](int gp) {
      const char* hr = smem + H_OFF + (gp & 1) * H_BUF + (s2_row0 >> 4) * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < SMI; ++i) ha[i] = ff_frag(hr + i * 1024);
      w2b = smem + W2_OFF + (gp & 1) * W2_CHUNK + (s2_col0 >> 4) * 1024 + lane * 16;
    };
    auto stage2_group = [&](int jg) {
#pragma unroll
      for (int jj = 0; jj < SNJ / 5; ++jj) {
        const int j = jg * (SNJ / 5) + jj;
        const uint4 bw = ff_frag(w2b + j * 1024);
#pragma unroll
        for (int i = 0; i < SMI; ++i) {
          if (FABL(8)) {
            asm volatile("" ::"v"(bw.x), "v"(ha[i].x));
            continue;
          }
          Y[i][j] = mfma16(bw, ha[i], Y[i][j]);
        }
      }
    };

    for (int c = 0; c < FF_CHUNKS; ++c, ++g) {
      // W1(g) [issued during iteration g - 1] and W2(g - 1) landed; h(g - 1) written by every wave; everyone is done with
      // the buffers this iteration's copies refill (W1 ring slot of g - 1, W2 ring slot of g - 2).  The weights are
      // L2-resident (2.4 MB, every CU streams the same bytes), so one chunk of lookahead covers the copies.
      if (!FABL(128)) {
        ff_wait_vm<0>();
        ff_barrier();
      }
      stream_setup(g);
      const char* w1b = smem + W1_OFF + (g & 1) * W1_CHUNK + (2 * wn) * 1024 + lane * 16;
      f32x4_t P[MI][2];
#pragma unroll
      for (int i = 0; i < MI; ++i) P[i][0] = P[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const bool prev = c > 0;
      char* hb = smem + H_OFF + (g & 1) * H_BUF;
      auto ln_fold = [&](int jj) {
        if (FABL(4)) return;
        const int col = 64 * c + 16 * (2 * wn + jj) + 4 * lq;      // interleaved W1 row of this lane's 4 columns
        const float4 b4 = *reinterpret_cast<const float4*>(tab + col);
        const float4 s4 = *reinterpret_cast<const float4*>(tab + 2 * FF_H + col);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          P[i][jj][0] = fmaf(rs[i], P[i][jj][0], fmaf(rm[i], s4.x, b4.x));
          P[i][jj][1] = fmaf(rs[i], P[i][jj][1], fmaf(rm[i], s4.y, b4.y));
          P[i][jj][2] = fmaf(rs[i], P[i][jj][2], fmaf(rm[i], s4.z, b4.z));
          P[i][jj][3] = fmaf(rs[i], P[i][jj][3], fmaf(rm[i], s4.w, b4.w));
        }
      };
      auto gelu_block = [&](int jj, int ip) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // lanes 0-31 hold the value columns, lanes 32-63 the gate columns of a 16-column block: after the swap lanes
          // 0-31 have value AND gate of row block 2 ip, lanes 32-63 those of row block 2 ip + 1
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[2 * ip][jj][r]),
                                                     __float_as_uint(P[2 * ip + 1][jj][r]), false, false);
          o[r] = __uint_as_float(sw[0]) * (FABL(2) ? __uint_as_float(sw[1]) : gelu_f(__uint_as_float(sw[1])));
        }
        const int I = MI * wm + 2 * ip + (lane >> 5);                // 16-row block of the tile
        if (FABL(32)) {
          asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
          return;
        }
        *reinterpret_cast<uint2*>(hb + I * 1024 + ((2 * wn + jj) * 16 + lrow) * 16 + (lq & 1) * 8) =
            make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
      };
      auto s1_mfma = [&](int jj, int ks, const uint4& bfrag) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (FABL(16)) {
            asm volatile("" ::"v"(bfrag.x), "v"(xa[i][ks].x));
            continue;
          }
          P[i][jj] = mfma16(bfrag, xa[i][ks], P[i][jj]);
        }
      };
      // Three MFMA segments per iteration, each carrying one share of the other work (with one wave per SIMD VALU work only
      // overlaps the matrix pipe when it sits BETWEEN this wave's own MFMAs; with two the partner wave helps as well):
      //   (a) columns 0-15 of P over the 10 k-steps            + the weight copies of this iteration
      //   (b) columns 16-31 of P                               + LayerNorm fold / GEGLU / h write of columns 0-15
      //   (c) stage 2 of the previous chunk                    + LayerNorm fold / GEGLU / h write of columns 16-31
      {
        constexpr int PF = VX_FF_PF;
        uint4 bq[PF + 1];      // W1 fragments of column block 0, read PF k-steps ahead of their MFMAs
#pragma unroll
        for (int d = 0; d < PF; ++d) bq[d] = ff_frag(w1b + d * 4096);
#pragma unroll
        for (int ks = 0; ks < FF_KS; ++ks) {
          if (ks + PF < FF_KS) bq[(ks + PF) % (PF + 1)] = ff_frag(w1b + (ks + PF) * 4096);
          if (NW == 4) {
            issue_w1(ks);                            // one copy of W1(g + 1) per k-step ...
            if ((ks & 1) == 0) issue_w2(ks >> 1);    // ... and one of W2(g) every other
          } else {
            if ((ks & 1) == 0) issue_w1(ks >> 1);    // five copies of W1(g + 1) per wave ...
            else if ((ks >> 1) < W2_PER_WAVE) issue_w2(ks >> 1);   // ... and up to three of W2(g)
          }
          s1_mfma(0, ks, bq[ks % (PF + 1)]);
#ifndef VX_FF_NOSB
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      {
        constexpr int PF = VX_FF_PF;
        uint4 bq[PF + 1];
#pragma unroll
        for (int d = 0; d < PF; ++d) bq[d] = ff_frag(w1b + d * 4096 + 1024);
#pragma unroll
        for (int ks = 0; ks < FF_KS; ++ks) {
          if (ks + PF < FF_KS) bq[(ks + PF) % (PF + 1)] = ff_frag(w1b + (ks + PF) * 4096 + 1024);
          s1_mfma(1, ks, bq[ks % (PF + 1)]);
          if (ks == 0) ln_fold(0);
          if (ks == 3) gelu_block(0, 0);
          if (MI == 4 && ks == 6) gelu_block(0, MI == 4 ? 1 : 0);
        }
      }
      if (prev) stage2_begin(g - 1);
      ln_fold(1);
      if (prev) stage2_group(0);
      if (prev) stage2_group(1);
      gelu_block(1, 0);
      if (prev) stage2_group(2);
      if (prev) stage2_group(3);
      if (MI == 4) gelu_block(1, MI == 4 ? 1 : 0);
      if (prev) stage2_group(4);
    }
    // ---- the next tile's x fragments (the registers are free from here on), then stage 2 of the tile's last chunk
    // (the copy wait comes FIRST: vmcnt retires in order, a wait behind the x loads would wait for them as well)
    if (!FABL(128)) ff_wait_vm<0>();
    if (XPF && ti + 1 < my_tiles) load_x(ti + 1);
    if (!FABL(128)) ff_barrier();
    stage2_begin(g - 1);
#pragma unroll
    for (int jg = 0; jg < 5; ++jg) stage2_group(jg);

    // ---------------------------------------------------- epilogue: out = x + Y + b2   (8-byte stores, C^T fragments)
    const bf16_t* __restrict__ res = (const bf16_t*)p.residual;
    bf16_t* __restrict__ out = (bf16_t*)p.out;
    const int m0 = tile0 + s2_row0;
    // all residual loads of the lane first, then the arithmetic and the stores - a load -> use -> store chain per item
    // would drain the store queue at every step (vmcnt counts stores too)
    uint2 rv[SMI][SNJ];
#pragma unroll
    for (int j = 0; j < SNJ; ++j)
#pragma unroll
      for (int i = 0; i < SMI; ++i)
        rv[i][j] = *reinterpret_cast<const uint2*>(res + (size_t)(m0 + 16 * i + lrow) * p.ldr + s2_col0 + 16 * j + 4 * lq);
#pragma unroll
    for (int j = 0; j < SNJ; ++j) {
      const int col = s2_col0 + 16 * j + 4 * lq;
      const float4 b4 = *reinterpret_cast<const float4*>(tab + 4 * FF_H + col);
#pragma unroll
      for (int i = 0; i < SMI; ++i) {
        const size_t row = (size_t)(m0 + 16 * i + lrow);
        const uint2 r2 = rv[i][j];
        const float v0 = Y[i][j][0] + b4.x + e16_lo(r2.x);
        const float v1 = Y[i][j][1] + b4.y + e16_hi(r2.x);
        const float v2 = Y[i][j][2] + b4.z + e16_lo(r2.y);
        const float v3 = Y[i][j][3] + b4.w + e16_hi(r2.y);
        *reinterpret_cast<uint2*>(out + row * p.ldo + col) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
  ff_wait_vm<0>();   // the run-ahead copies of the weight stream must land before this workgroup's LDS is handed on
}

// W1i [2 H, C] (value / gate rows interleaved in blocks of 8, LayerNorm-folded) -> [chunk][k-step][column block][lane][8]
// W2  [C, H]                                                                     -> [chunk][column block][lane][8]
__global__ void ff_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, bf16_t* __restrict__ w1t,
                               bf16_t* __restrict__ w2t) {
  const int n1 = 2 * FF_H * FF_C / 8, n2 = FF_C * FF_H / 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n1 + n2; idx += gridDim.x * blockDim.x) {
    if (idx < n1) {
      const int lane = idx & 63, j = (idx >> 6) & 3, ks = (idx >> 8) % FF_KS, c = idx / (256 * FF_KS);
      const bf16_t* src = w1 + (size_t)(64 * c + 16 * j + (lane & 15)) * FF_C + 32 * ks + 8 * (lane >> 4);
      *reinterpret_cast<uint4*>(w1t + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(src);
    } else {
      const int k = idx - n1;
      const int lane = k & 63, j = (k >> 6) % 20, c = k / (64 * 20);
      const bf16_t* src = w2 + (size_t)(16 * j + (lane & 15)) * FF_H + 32 * c + 8 * (lane >> 4);
      *reinterpret_cast<uint4*>(w2t + (size_t)k * 8) = *reinterpret_cast<const uint4*>(src);
    }
  }
}

}  // namespace

extern "C" int vx_ff_pack_weights(const void* w1_interleaved, const void* w2, void* w1t, void* w2t, int c, int hidden,
                                  void* stream_) {
  VX_REQUIRE(w1_interleaved != nullptr && w2 != nullptr && w1t != nullptr && w2t != nullptr, "vx_ff_pack_weights: null pointer");
  VX_REQUIRE(c == FF_C && hidden == FF_H, "vx_ff_pack_weights: only C = %d, hidden = %d (the 64x64 level) is built", FF_C, FF_H);
  hipLaunchKernelGGL(ff_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)w1_interleaved,
                     (const bf16_t*)w2, (bf16_t*)w1t, (bf16_t*)w2t);
  return vx_check_launch("vx_ff_pack_weights");
}

extern "C" int vx_ff_fused(const vx_ff_params* pp, void* stream_) {
  const vx_ff_params& p = *pp;
  VX_REQUIRE(p.x != nullptr && p.w1t != nullptr && p.w2t != nullptr && p.out != nullptr && p.ln_stats != nullptr &&
                 p.ln_colsum != nullptr && p.residual != nullptr, "vx_ff_fused: null pointer");
  VX_REQUIRE(p.c == FF_C && p.hidden == FF_H, "vx_ff_fused: only C = %d, hidden = %d (the 64x64 level) is built", FF_C, FF_H);
  VX_REQUIRE(p.m > 0 && (p.m % FF_BM) == 0, "vx_ff_fused: m=%d must be a multiple of %d", p.m, FF_BM);
  VX_REQUIRE((p.ldx % 8) == 0 && (p.ldo % 4) == 0 && (p.ldr % 4) == 0, "vx_ff_fused: row strides");
  static bool attr_set = false;
  static int cus = 256, four_waves = 0;
  if (!attr_set) {
    // VX_FF_WAVES=4 (A/B knob): the four-wave, one-per-SIMD form; default: eight waves
    const char* ev = getenv("VX_FF_WAVES");
    four_waves = ev && atoi(ev) == 4;
    hipError_t e = hipFuncSetAttribute((const void*)ff_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)ff_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
    if (e != hipSuccess) {
      vx_set_error("vx_ff_fused: hipFuncSetAttribute(%d B LDS) failed: %s", FF_LDS, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    attr_set = true;
  }
  const int tiles = p.m / FF_BM;
  g_vx_last_kernel = four_waves ? "ff_fused_kernel<4>" : "ff_fused_kernel<2>";
  if (four_waves)
    hipLaunchKernelGGL(ff_fused_kernel<4>, dim3(tiles < cus ? tiles : cus), dim3(256), FF_LDS, (hipStream_t)stream_, p);
  else
    hipLaunchKernelGGL(ff_fused_kernel<2>, dim3(tiles < cus ? tiles : cus), dim3(512), FF_LDS, (hipStream_t)stream_, p);
  return vx_check_launch("vx_ff_fused");
}
