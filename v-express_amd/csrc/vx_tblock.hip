// Fused temporal self-attention block of the 64x64 level (C = 320, 8 heads of 40, 16 frames) on gfx950:
//   x <- x + to_out( softmax_over_frames( q k^T / sqrt(d) ) v ),   [q | k | v] = (LN(x) + pe[frame]) Wqkv^T + b
// in ONE launch (VersatileAttention.forward inside TemporalTransformerBlock.forward, modules/motion_module.py:243-256,
// 351-388; until now vx_gemm (LayerNorm-folded QKV, 84 MB in / 252 MB out at the 64x64 level) + vx_temporal_attention
// (252 MB in / 84 MB out) + vx_gemm (out-projection + residual, 168 MB in / 84 MB out): 924 MB of traffic for an operation
// whose inputs and outputs are 168 MB).  Built from the fused feed-forward's parts (vx_ff.hip):
//   * a tile = 8 consecutive pixels x their 16 frames = 128 rows, one 16-row MFMA block per PIXEL (its 16 frames), so
//     that attention over the frame axis is a 16 x 16 problem that lives inside one wave's registers;
//   * phase 1, waves as (pixel groups) x 2 (the heads of a head pair): a wave's x rows stay in registers as MFMA
//     fragments for the whole tile; the weights are pre-tiled (vx_tblock_pack) into fragment-major 20 KB chunks = one
//     column block of 16 for each of the two heads of a pair, streamed through a three-slot LDS ring by LDS-DMA two chunks
//     ahead, one barrier per chunk; 8 blocks per head: Q0 Q1 K0 K1 M V0 V1 V2  (M = q32..39 | k32..39, V2 = v32..39 | 8
//     zero rows: 128 columns per head instead of 120);
//   * Q, K come out of the MFMA TRANSPOSED (A = weights, B = x: lane = frame, 4 channels), V plain (A = x, B = weights:
//     lane = channel, 4 frames) - exactly the operand layouts the 16x16 attention wants: S^T = K Q^T takes the packed
//     accumulators of K and Q as its A and B operands (the K dimension of an MFMA may be permuted freely as long as both
//     operands agree), P^T = softmax(S^T) is the B operand of O^T = V^T P^T (v_mfma_f32_16x16x16_bf16) with V's packed
//     accumulators as A.  Nothing of the attention touches LDS or memory;
//   * O^T (bf16) of all heads is parked in 80 KB of LDS in the out-projection's B-fragment order (16-byte linear writes);
//   * phase 2, waves as 2 (pixel groups) x (column groups): Y = O Wo^T (K = 80 per head pair as 32 + 32 + 16), the
//     out-projection weights run through the same ring as three parts per head pair, while the next tile's x rows load.
//     (Accumulating Y beside phase 1 - the feed-forward kernel's structure - needs x + Y + the packed q / k / v of the
//     head in flight + accumulators + operands = ~265 of 256 registers with eight waves, ~480 of 512 with four: the first
//     build spilled 320 / 407 registers.  In two phases the peak is x 80 + q/k/v 32 + 30 resp. Y 80 + x 80.)
//   * LayerNorm statistics come from the x rows the wave holds (two-pass, in registers) unless the caller passes them.
// Round 5: F = 24 frames (the reference's default window, inference.py:67) as well.  A pixel's 24 frames are TWO 16-row MFMA
// blocks (frames 0-15, frames 16-23 + 8 padding rows that re-read frame 23): a tile = 4 pixels x 2 blocks = the same 8
// blocks, a wave holds both blocks of its pixel, the attention of a (pixel, head) is 2 x 2 score blocks with the padded
// keys masked to -inf (their probabilities are exactly 0) and the padded query rows never stored; the bias / positional
// tables of a chunk grow to one 16 x 16 table per (head, block half).  25 % of the projection MFMAs multiply padding.
// Rounding points are those of the three launches (q, k, v, P, O rounded to bf16; fp32 accumulation in k order; the
// LayerNorm fold as two FMAs), so the result differs from theirs only by the summation order inside the 16 x 16 products.
#include "vx_common.h"
#include "vx_gemm_common.h"
#include "../../include/vexpress_hip.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int TB_C = 320, TB_HEADS = 8, TB_D = 40;
#ifndef VX_TB_WAVES
#define VX_TB_WAVES 8
#endif
constexpr int TB_NW = VX_TB_WAVES, TB_NPX = 16 / TB_NW;   // waves per workgroup (8 or 4), pixels per wave in phase 1
constexpr int TB_KS = TB_C / 32;                 // 10 k-steps of 32
constexpr int TB_PCOLS = TB_HEADS * 8 * 16;      // 1024 packed columns: 8 blocks of 16 per head
constexpr int TB_WBYTES = 2 * 16 * TB_C * 2;     // 20480 B of weights per QKV chunk: [k-step 10][head of the pair 2][lane 64][16 B]
constexpr int TB_QKV_CHUNKS = 32, TB_TILE_CHUNKS = 44;   // + 4 head pairs x 3 out-projection parts
constexpr int WO_P01 = 20 * 1024, WO_P2 = 20 * 512;
constexpr int WO_PAIR = 2 * WO_P01 + WO_P2;      // 51200 B per head pair
constexpr int RING_OFF = 0;                      // 3 slots
constexpr int TB_BLOCKS = 8;                     // 16-row MFMA blocks per tile
constexpr int O_PIX = 2560, O_PAIR = TB_BLOCKS * O_PIX;  // O^T of a block: [kb0 1024 | kb1 1024 | kb2 512]; 20480 per head pair
// geometry that depends on the window length F (16 or 24 frames): HB blocks per pixel, PIX pixels per tile
template <int F>
struct TbGeo {
  static_assert(F == 16 || F == 24, "vx_tblock: 16 or 24 frames");
  static constexpr int HB = F > 16 ? 2 : 1;
  static constexpr int PIX = TB_BLOCKS / HB;
  static constexpr int NBLK = 20 + 2 * HB;              // 1-KiB copy blocks of a QKV chunk: 20 of weights + the tables
  static constexpr int SLOT = TB_WBYTES + 2 * HB * 1024;  // + bias / positional tables, 16 x 16 fp32 per (head, block half): 22528 / 24576 B
  static constexpr int O_OFF = 3 * SLOT;                // O^T of the tile, [head pair][block]
  static constexpr int CS_OFF = O_OFF + 4 * O_PAIR;     // column sums of the folded weight, packed column order (fp32)
  static constexpr int BO_OFF = CS_OFF + TB_PCOLS * 4;  // out-projection bias (fp32)
  static constexpr int ST_OFF = BO_OFF + TB_C * 4;      // (rstd, -mean rstd) of the tile's rows, [block][row]
  static constexpr int LDS = ST_OFF + TB_BLOCKS * 16 * 8;   // 155904 / 162048 <= 163840
};

// Compile-time ablation switches (tools/build_tb_variants.sh; never defined for the product library):
//   1 no weight copies after the prologue   2 no attention (O = the V rows)   4 no LayerNorm fold / tables
//   8 no out-projection MFMAs   16 no QKV MFMAs   64 no LDS fragment reads   128 no barrier / copy wait
#ifdef VX_TB_ABLATE
#define TABL(bit) (((VX_TB_ABLATE) & (bit)) != 0)
#else
#define TABL(bit) false
#endif
#ifndef VX_TB_PF
#define VX_TB_PF 2
#endif

typedef vx_e16x4_t tb_s4;
__device__ __forceinline__ f32x4_t mfma16k(const uint2& a, const uint2& b, f32x4_t c) {   // 16 x 16 x 16
  return VX_MFMA_16x16x16(__builtin_bit_cast(tb_s4, a), __builtin_bit_cast(tb_s4, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 tb_frag(const char* p) {
  if (TABL(64)) return make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ uint2 tb_frag8(const char* p) {
  if (TABL(64)) return make_uint2(0x3c003c00u, 0x3c003c00u);
  return *reinterpret_cast<const uint2*>(p);
}
template <int N>
__device__ __forceinline__ void tb_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tb_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// packed column (head, block, row) -> column of the [3 C] q | k | v projection, or -1 (zero padding)
//   block 0 / 1: q 0..15 / 16..31   2 / 3: k   4: q 32..39 | k 32..39   5 / 6: v 0..15 / 16..31   7: v 32..39 | zero
__host__ __device__ inline int tb_src_col(int head, int blk, int r) {
  const int q = head * TB_D, k = TB_C + head * TB_D, v = 2 * TB_C + head * TB_D;
  switch (blk) {
    case 0: return q + r;
    case 1: return q + 16 + r;
    case 2: return k + r;
    case 3: return k + 16 + r;
    case 4: return r < 8 ? q + 32 + r : k + 32 + (r - 8);
    case 5: return v + r;
    case 6: return v + 16 + r;
    default: return r < 8 ? v + 32 + r : -1;
  }
}

template <int F>
__global__ __launch_bounds__(64 * TB_NW, TB_NW == 8 ? 2 : 1) void tblock_kernel(const vx_tblock_params p,
                                                                                 const float scale_log2e) {
  using G_ = TbGeo<F>;
  constexpr int HB = G_::HB, TB_PIX = G_::PIX, TB_SLOT = G_::SLOT, NBLK = G_::NBLK;
  constexpr int O_OFF = G_::O_OFF, CS_OFF = G_::CS_OFF, BO_OFF = G_::BO_OFF, ST_OFF = G_::ST_OFF;
  constexpr int NW = TB_NW, NT = 64 * NW, NPX = TB_NPX;
  static_assert(HB == 1 || NPX == 2, "F = 24: a wave must hold both blocks of its pixel");
  constexpr int CPW = (NBLK + NW - 1) / NW;       // copies per wave and chunk (slots past the end repeat a block)
  constexpr int SNJ = 20 / (NW / 2);              // out-projection column blocks per wave
  constexpr int NX = NPX * TB_KS;                 // loads of one wave's x rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;        // phase 1: pixels NPX wm .. + NPX - 1; head wn of the pair
  const int lrow = lane & 15, lq = lane >> 4;
  const uint32_t lds0 = lds_addr_of(smem);
  const uint32_t lane16 = (uint32_t)lane * 16u;

  {
    float* cs = reinterpret_cast<float*>(smem + CS_OFF);
    for (int i = tid; i < TB_PCOLS; i += NT) cs[i] = p.colsum_p[i];
    float* bo = reinterpret_cast<float*>(smem + BO_OFF);
    for (int i = tid; i < TB_C; i += NT) bo[i] = p.bias_o != nullptr ? p.bias_o[i] : 0.f;
  }
  __syncthreads();

  const int tiles_per_item = p.hw / TB_PIX;
  const int n_tiles = p.b * tiles_per_item;
  const int G = gridDim.x;
  const int tfirst = (int)blockIdx.x;
  const int my_tiles = (n_tiles - tfirst + G - 1) / G;
  const size_t frame_stride = (size_t)p.hw * p.ldx;      // elements between frame f and f + 1 of a pixel
  auto tile_row0 = [&](int ti) {                         // row of (frame 0, first pixel) of the ti-th tile of this block
    const int t = tfirst + ti * G;
    const int bb = t / tiles_per_item, px = (t - bb * tiles_per_item) * TB_PIX;
    return (size_t)bb * F * p.hw + px;
  };

  const char* __restrict__ wq = (const char*)p.wqkv_t;
  const char* __restrict__ wo = (const char*)p.wo_t;
  const bf16_t* __restrict__ x = (const bf16_t*)p.x;
  const float2* __restrict__ st_in = reinterpret_cast<const float2*>(p.ln_stats);
  const float* cs_tab = reinterpret_cast<const float*>(smem + CS_OFF);
  const float* bo_tab = reinterpret_cast<const float*>(smem + BO_OFF);

  // ---- x rows of the wave: lane = frame lrow of pixel NPX wm + i, k group lq (second MFMA operand of Q / K, first of V)
  // block NPX wm + i of the tile = (pixel, block half): F = 16: (NPX wm + i, 0); F = 24: (wm, i) - rows 16 half + lrow of the
  // pixel's frames, rows past the last frame re-read it (padding: masked as keys, never stored as queries)
  auto blk_pix = [&](int b) { return HB == 1 ? b : b >> 1; };
  auto blk_frame = [&](int b, int r) {
    const int fr = HB == 1 ? r : 16 * (b & 1) + r;
    return fr < F ? fr : F - 1;
  };
  uint4 xa[NPX][TB_KS];
  auto load_x = [&](int ti) {
    const size_t r0 = tile_row0(ti);
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const bf16_t* row = x + (r0 + blk_pix(NPX * wm + i)) * p.ldx + blk_frame(NPX * wm + i, lrow) * frame_stride + 8 * lq;
#pragma unroll
      for (int ks = 0; ks < TB_KS; ++ks) xa[i][ks] = *reinterpret_cast<const uint4*>(row + 32 * ks);
    }
  };
  load_x(0);       // BEFORE the first copies: vmcnt retires in order, the counted waits below assume copies are youngest

  // ---- weight stream: 44 chunks per tile (32 QKV chunks = (head pair, block), then 12 out-projection parts), the same
  // for every tile.  Ring slot of the chunk in flight = a running counter mod 3; during an iteration the chunk TWO ahead
  // is copied (its slot held the previous chunk, which every wave left at this iteration's barrier).  Every wave issues the
  // same number of copies per chunk (CPW; slots past the end repeat a block: same bytes, same place): the counted
  // waits are wave-independent.
  bool in_loop = false;
  auto issue = [&](int c, int slot, int q) {        // q-th copy of this wave for chunk c (0 .. 43) into ring slot `slot`
    if (TABL(1) && in_loop) return;
    const char* src;
    int nblk = NBLK;
    if (c < TB_QKV_CHUNKS) {
      src = wq + (size_t)c * TB_SLOT;
    } else {
      nblk = 20;
      const int hp2 = (c - TB_QKV_CHUNKS) / 3, part = (c - TB_QKV_CHUNKS) - 3 * hp2;
      src = wo + (size_t)hp2 * WO_PAIR + part * WO_P01;
      if (part == 2) nblk = 10;
    }
    const int iq = wave + NW * q;                                   // constant divisors: no scalar division sequence
    const int blk = nblk == 10 ? iq % 10 : (nblk == 20 ? iq % 20 : iq % NBLK);
    glds16_s(src + blk * 1024, lane16, lds0 + RING_OFF + slot * TB_SLOT + blk * 1024);
  };
  {
#pragma unroll
    for (int q = 0; q < CPW; ++q) issue(0, 0, q);
#pragma unroll
    for (int q = 0; q < CPW; ++q) issue(1, 1, q);
  }
  in_loop = true;
  int slot = 0;                                    // ring slot of the current chunk
  auto next2 = [&](int c) { return c + 2 < TB_TILE_CHUNKS ? c + 2 : c + 2 - TB_TILE_CHUNKS; };
  auto slot2 = [&]() { return slot == 0 ? 2 : slot - 1; };           // (slot + 2) % 3
  auto advance = [&]() { slot = slot == 2 ? 0 : slot + 1; };

  // phase 2 / epilogue ownership: 2 (pixel groups of 4) x NW / 2 (column groups of 16 SNJ)
  const int s2_pix0 = 4 * (wave / (NW / 2)), s2_cg = wave % (NW / 2), s2_col0 = 16 * SNJ * s2_cg;

  for (int ti = 0; ti < my_tiles; ++ti) {
    const size_t row0 = tile_row0(ti);
    // ---- LayerNorm scalars of the wave's rows: (rstd, -mean rstd); lane = frame lrow of pixel NPX wm + i
    float rs[NPX], rm[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      if (st_in != nullptr) {
        const float2 t = st_in[row0 + blk_pix(NPX * wm + i) + (size_t)blk_frame(NPX * wm + i, lrow) * p.hw];
        rs[i] = t.y;
        rm[i] = -t.x * t.y;
      } else {
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < TB_KS; ++ks) {
          float v[8];
          unpack_bf16x8(xa[i][ks], v);
          sm += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        sm = wave_xor_sum(sm, 16);
        sm = wave_xor_sum(sm, 32);
        const float mean = sm * (1.0f / TB_C);
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < TB_KS; ++ks) {
          float v[8];
          unpack_bf16x8(xa[i][ks], v);
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[e] - mean;
            t += d * d;
          }
          sq += t;
        }
        sq = wave_xor_sum(sq, 16);
        sq = wave_xor_sum(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / TB_C) + p.ln_eps);
        rs[i] = rstd;
        rm[i] = -mean * rstd;
      }
      if (TABL(4)) {
        rs[i] = 1.f;
        rm[i] = 0.f;
      }
      // the V blocks need the scalars of frames 4 lq .. 4 lq + 3 (their accumulator rows): [pixel][frame] table in LDS.
      // Both waves of a pixel group write the same values; the first reader is five barriers away.
      if (lq == 0)
        *reinterpret_cast<float2*>(smem + ST_OFF + ((NPX * wm + i) * 16 + lrow) * 8) = make_float2(rs[i], rm[i]);
    }

    // ======================================================================= phase 1: q, k, v and the attention, per head pair
    for (int hp = 0; hp < 4; ++hp) {
      const int head = 2 * hp + wn;
      uint2 Qp[NPX][2], Kp[NPX][2], Mp[NPX], Vp[NPX][3];      // packed bf16 results of this wave's head and pixels

      auto chunk = [&](auto blk_c) {
        constexpr int blk = decltype(blk_c)::value;
        constexpr bool plain = blk >= 5;                       // V blocks: A = x, B = weights
        const int c = 8 * hp + blk;
        // chunk c has landed (issued two iterations ago; the copies of chunk c + 1 may still be in flight) and every
        // wave is done with the slot of chunk c - 1
        if (!TABL(128)) {
          tb_wait_vm<CPW>();
          tb_barrier();
        }
        const int pc = (head * 8 + blk) * 16;
        const char* wb = smem + RING_OFF + slot * TB_SLOT + wn * 1024 + lane * 16;
        // bias + positional row of the block's 16 columns, travelling with the chunk: [frame][column] for the transposed
        // blocks, [column][frame] for the V blocks - either way this lane's four values sit at [lrow][4 lq .. + 3]
        // (F = 24: one table per block half - the half = the wave's block index i)
        float4 tqh[HB];
#pragma unroll
        for (int hh = 0; hh < HB; ++hh) {
          tqh[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!TABL(4))
            tqh[hh] = *reinterpret_cast<const float4*>(smem + RING_OFF + slot * TB_SLOT + TB_WBYTES + (wn * HB + hh) * 1024 +
                                                       (lrow * 16 + 4 * lq) * 4);
        }
        const int cn = next2(c), sn = slot2();
        f32x4_t P[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) P[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
          constexpr int PF = VX_TB_PF;
          uint4 bq[PF + 1];
#pragma unroll
          for (int d = 0; d < PF; ++d) bq[d] = tb_frag(wb + d * 2048);
#pragma unroll
          for (int ks = 0; ks < TB_KS; ++ks) {
            if (ks + PF < TB_KS) bq[(ks + PF) % (PF + 1)] = tb_frag(wb + (ks + PF) * 2048);
            // the copies of chunk c + 2, spread over the k-steps (a burst stalls the issuing wave)
            if ((ks & 1) == 0 && (ks >> 1) < CPW) issue(cn, sn, ks >> 1);
            const uint4& wf = bq[ks % (PF + 1)];
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
              if (TABL(16)) {
                asm volatile("" ::"v"(wf.x), "v"(xa[i][ks].x));
                continue;
              }
              P[i] = plain ? mfma16(xa[i][ks], wf, P[i]) : mfma16(wf, xa[i][ks], P[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // bias / LayerNorm fold / rounding into the block's packed home
        if constexpr (blk < 5) {
          const float4 s4 = *reinterpret_cast<const float4*>(cs_tab + pc + 4 * lq);
#pragma unroll
          for (int i = 0; i < NPX; ++i) {
            const float4 tq = tqh[HB == 1 ? 0 : i];
            const float v0 = fmaf(rs[i], P[i][0], fmaf(rm[i], s4.x, tq.x));
            const float v1 = fmaf(rs[i], P[i][1], fmaf(rm[i], s4.y, tq.y));
            const float v2 = fmaf(rs[i], P[i][2], fmaf(rm[i], s4.z, tq.z));
            const float v3 = fmaf(rs[i], P[i][3], fmaf(rm[i], s4.w, tq.w));
            uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            // pinned here: otherwise the compiler sinks the fold of one of the pixels down to the attention phase and keeps
            // accumulators, table values and column sums of five blocks alive instead of two packed registers each
            asm volatile("" : "+v"(pk.x), "+v"(pk.y));
            if constexpr (blk < 2) Qp[i][blk] = pk;
            else if constexpr (blk < 4) Kp[i][blk - 2] = pk;
            else Mp[i] = pk;
          }
        } else {
          const float s1 = cs_tab[pc + lrow];
#pragma unroll
          for (int i = 0; i < NPX; ++i) {
            // (rstd, -mean rstd) of frames 4 lq .. 4 lq + 3 of pixel NPX wm + i
            const float4 tq = tqh[HB == 1 ? 0 : i];
            const float4 a = *reinterpret_cast<const float4*>(smem + ST_OFF + ((NPX * wm + i) * 16 + 4 * lq) * 8);
            const float4 b = *reinterpret_cast<const float4*>(smem + ST_OFF + ((NPX * wm + i) * 16 + 4 * lq + 2) * 8);
            const float v0 = fmaf(a.x, P[i][0], fmaf(a.y, s1, tq.x));
            const float v1 = fmaf(a.z, P[i][1], fmaf(a.w, s1, tq.y));
            const float v2 = fmaf(b.x, P[i][2], fmaf(b.y, s1, tq.z));
            const float v3 = fmaf(b.z, P[i][3], fmaf(b.w, s1, tq.w));
            uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            asm volatile("" : "+v"(pk.x), "+v"(pk.y));
            Vp[i][blk - 5] = pk;
          }
        }
        advance();
      };
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});
      chunk(std::integral_constant<int, 2>{});
      chunk(std::integral_constant<int, 3>{});
      chunk(std::integral_constant<int, 4>{});
      chunk(std::integral_constant<int, 5>{});
      chunk(std::integral_constant<int, 6>{});
      chunk(std::integral_constant<int, 7>{});

      // ---- attention over the frames of (pixel, head): everything in this wave's registers.  Query block i against HB key
      // blocks: F = 16: its own; F = 24: both blocks of the pixel (keys 16 + 4 lq + r >= 24 are padding: -inf)
      // channels 32..39: the mixed block holds q in lanes 0-31 and k in lanes 32-63 -> (q | 0), (k | 0)
      uint2 mq[NPX], mk[NPX];
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        auto m0 = __builtin_amdgcn_permlane32_swap(Mp[i].x, 0u, false, false);
        auto m1 = __builtin_amdgcn_permlane32_swap(Mp[i].y, 0u, false, false);
        mq[i] = make_uint2(m0[0], m1[0]);
        mk[i] = make_uint2(m0[1], m1[1]);
      }
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        uint2 o01[2], o2;
        if (TABL(2)) {
          o01[0] = Vp[i][0];
          o01[1] = Vp[i][1];
          o2 = Vp[i][2];
        } else {
          f32x4_t sc[HB];
#pragma unroll
          for (int kbi = 0; kbi < HB; ++kbi) {
            const int kb = HB == 1 ? i : kbi;
            // S^T[key frame 16 kb + 4 lq + r][query row lrow] = sum_d K[key][d] Q[query][d]
            const f32x4_t s_main = mfma16(make_uint4(Kp[kb][0].x, Kp[kb][0].y, Kp[kb][1].x, Kp[kb][1].y),
                                          make_uint4(Qp[i][0].x, Qp[i][0].y, Qp[i][1].x, Qp[i][1].y),
                                          f32x4_t{0.f, 0.f, 0.f, 0.f});
            // Its own accumulator, added on the VALU - NOT chained through the C operand: a v_mfma_f32_16x16x16_bf16 that
            // takes the result of the v_mfma_f32_16x16x32_bf16 right in front of it as C read it too early on the hardware
            // (hipcc 7.2 puts no wait states between that pair; measured: the 32-channel term of S went missing for whichever
            // pixel had fewer than ~5 instructions between the two, run-to-run different for the late waves;
            // profiles/r04m_tblock_probes.txt).  VALU reads of MFMA results are interlocked by the compiler as everywhere.
            const f32x4_t s_mix = mfma16k(mk[kb], mq[i], f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[kbi][r] = s_main[r] + s_mix[r];
            if (HB == 2 && kbi == 1 && lq >= 2) {      // frames 24 .. 31 do not exist
#pragma unroll
              for (int r = 0; r < 4; ++r) sc[kbi][r] = -INFINITY;
            }
          }
          float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
          if (HB == 2) mx = fmaxf(mx, fmaxf(fmaxf(sc[HB - 1][0], sc[HB - 1][1]), fmaxf(sc[HB - 1][2], sc[HB - 1][3])));
          mx = wave_xor_max(mx, 16);
          mx = wave_xor_max(mx, 32);
          const float ms = mx * scale_log2e;
          float sum = 0.f;
          uint2 pb[HB];
#pragma unroll
          for (int kbi = 0; kbi < HB; ++kbi) {
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pr[r] = __builtin_amdgcn_exp2f(fmaf(sc[kbi][r], scale_log2e, -ms));
              sum += pr[r];
            }
            pb[kbi] = make_uint2(pack_bf16x2(pr[0], pr[1]), pack_bf16x2(pr[2], pr[3]));
          }
          sum = wave_xor_sum(sum, 16);
          sum = wave_xor_sum(sum, 32);
          const float inv_l = 1.0f / sum;
          // O^T[channel 4 lq + r of block vb][query row lrow] = sum_key V[key][channel] P[query][key]
          uint2 o[3];
#pragma unroll
          for (int vb = 0; vb < 3; ++vb) {
            f32x4_t a = mfma16k(Vp[HB == 1 ? i : 0][vb], pb[0], f32x4_t{0.f, 0.f, 0.f, 0.f});
            if (HB == 2) a = mfma16k(Vp[1][vb], pb[HB - 1], a);      // (same MFMA shape chained through C: fine)
            o[vb] = make_uint2(pack_bf16x2(a[0] * inv_l, a[1] * inv_l), pack_bf16x2(a[2] * inv_l, a[3] * inv_l));
          }
          o01[0] = o[0];
          o01[1] = o[1];
          o2 = o[2];
        }
        char* ob = smem + O_OFF + hp * O_PAIR + (NPX * wm + i) * O_PIX;
        *reinterpret_cast<uint4*>(ob + wn * 1024 + lane * 16) = make_uint4(o01[0].x, o01[0].y, o01[1].x, o01[1].y);
        if (lq < 2) *reinterpret_cast<uint2*>(ob + 2048 + ((2 * wn + lq) * 16 + lrow) * 8) = o2;
      }
    }

    // ======================================================================= phase 2: Y = O Wo^T, 12 parts; next tile's x rows
    // (the first part starts from a literal zero accumulator: a zero-initialised Y is loop-invariant, the compiler hoists
    // the 80 zero registers out of the tile loop and they stay allocated all through phase 1)
    f32x4_t Y[4][SNJ];
    const bool more = ti + 1 < my_tiles;
    for (int hp2 = 0; hp2 < 4; ++hp2) {
      auto part_iter = [&](auto part_c, auto first_c) {
        constexpr int part = decltype(part_c)::value;
        constexpr bool first = decltype(first_c)::value;
        const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int c = TB_QKV_CHUNKS + 3 * hp2 + part;
        // The x loads of the next tile (issued behind the copies of the FIRST part iteration) are younger than the copies
        // of chunk c + 1 in the two iterations that follow it: those wait for "all but the youngest CPW + NX".
        if (!TABL(128)) {
          if (hp2 == 0 && part > 0 && more) tb_wait_vm<CPW + NX>();
          else tb_wait_vm<CPW>();
          tb_barrier();
        }
        const int cn = next2(c), sn = slot2();
#pragma unroll
        for (int q = 0; q < CPW; ++q) issue(cn, sn, q);
        if constexpr (part == 0) {
          if (hp2 == 0 && more) load_x(ti + 1);
        }
        const char* ob = smem + O_OFF + hp2 * O_PAIR + s2_pix0 * O_PIX + (part < 2 ? part * 1024 + lane * 16 : 2048 + lane * 8);
        const char* wb = smem + RING_OFF + slot * TB_SLOT + (SNJ * s2_cg) * (part < 2 ? 1024 : 512) + lane * (part < 2 ? 16 : 8);
        if constexpr (part < 2) {
          uint4 oa[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) oa[i] = tb_frag(ob + i * O_PIX);
#pragma unroll
          for (int j = 0; j < SNJ; ++j) {
            const uint4 bw = tb_frag(wb + j * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (TABL(8)) {
                if (first) Y[i][j] = zero4;
                asm volatile("" ::"v"(bw.x), "v"(oa[i].x));
                continue;
              }
              Y[i][j] = mfma16(bw, oa[i], first ? zero4 : Y[i][j]);
            }
          }
        } else {
          uint2 oa[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) oa[i] = tb_frag8(ob + i * O_PIX);
#pragma unroll
          for (int j = 0; j < SNJ; ++j) {
            const uint2 bw = tb_frag8(wb + j * 512);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (TABL(8)) {
                asm volatile("" ::"v"(bw.x), "v"(oa[i].x));
                continue;
              }
              Y[i][j] = mfma16k(bw, oa[i], Y[i][j]);
            }
          }
        }
        advance();
      };
      if (hp2 == 0) part_iter(std::integral_constant<int, 0>{}, std::true_type{});
      else part_iter(std::integral_constant<int, 0>{}, std::false_type{});
      part_iter(std::integral_constant<int, 1>{}, std::false_type{});
      part_iter(std::integral_constant<int, 2>{}, std::false_type{});
    }

    // ---- x <- x + Y + bias: lane = frame lrow of pixel s2_pix0 + i, columns s2_col0 + 16 j + 4 lq .. + 3; in batches of
    // JB column blocks: all residual loads of a batch first, then its arithmetic and stores (a load -> use -> store
    // chain per item would drain the store queue at every step: vmcnt counts stores too)
    bf16_t* __restrict__ out = (bf16_t*)p.x;
    // row of (block s2_pix0 + i, lrow): F = 24: the padding rows of the second block half (frame >= 24) are read from the
    // last frame and never stored
    // (element offsets fit 32 bits: the tensor is < 4 G elements - checked by the launcher)
    uint32_t erow[4];
    bool estore[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = s2_pix0 + i;
      erow[i] = (uint32_t)((row0 + blk_pix(b)) * p.ldx + (size_t)blk_frame(b, lrow) * frame_stride);
      estore[i] = HB == 1 || 16 * (b & 1) + lrow < F;
    }
    float so_s[4] = {0.f, 0.f, 0.f, 0.f}, so_q[4] = {0.f, 0.f, 0.f, 0.f};   // stats_out: this lane's share of row (pixel i, lrow)
    constexpr int JB = 3;       // column blocks per batch (rv = 4 JB uint2: the epilogue is the register peak of the kernel)
#pragma unroll
    for (int jb = 0; jb < SNJ; jb += JB) {
      uint2 rv[4][JB];
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (jb + j < SNJ)
            rv[i][j] = *reinterpret_cast<const uint2*>(out + erow[i] + s2_col0 + 16 * (jb + j) + 4 * lq);
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        if (jb + j >= SNJ) continue;
        const int col = s2_col0 + 16 * (jb + j) + 4 * lq;
        const float4 b4 = *reinterpret_cast<const float4*>(bo_tab + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint2 r2 = rv[i][j];
          const float v0 = Y[i][jb + j][0] + b4.x + e16_lo(r2.x);
          const float v1 = Y[i][jb + j][1] + b4.y + e16_hi(r2.x);
          const float v2 = Y[i][jb + j][2] + b4.z + e16_lo(r2.y);
          const float v3 = Y[i][jb + j][3] + b4.w + e16_hi(r2.y);
          const uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          if (estore[i]) *reinterpret_cast<uint2*>(out + erow[i] + col) = pk;
          if (p.stats_out != nullptr) {        // of the STORED values, as vx_row_stats would read them back
            const float r0 = e16_lo(pk.x), r1 = e16_hi(pk.x);
            const float r2_ = e16_lo(pk.y), r3 = e16_hi(pk.y);
            so_s[i] += (r0 + r1) + (r2_ + r3);
            so_q[i] = fmaf(r0, r0, so_q[i]); so_q[i] = fmaf(r1, r1, so_q[i]);
            so_q[i] = fmaf(r2_, r2_, so_q[i]); so_q[i] = fmaf(r3, r3, so_q[i]);
          }
        }
      }
    }
    // ---- stats_out: (mean, rstd) of the rows just written, for the LayerNorm fold of the NEXT consumer (the second
    // attention block, the feed-forward).  A row's 320 columns sit in 4 lanes x NW / 2 waves: lanes fold over lq, waves
    // meet in LDS - in the O^T region of head pair 0, which nobody has read since the third iteration of phase 2 and nobody
    // writes before the eighth barrier of the next tile - one barrier, then wave w finishes rows 16 w .. 16 w + 15.  One-pass
    // variance (E[x^2] - mean^2 from float32 sums in a fixed order), as the GEMM epilogue's row_stats_out.
    if (p.stats_out != nullptr) {
      float2* scr = reinterpret_cast<float2*>(smem + O_OFF);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = so_s[i], b = so_q[i];
        a = wave_xor_sum(a, 16); a = wave_xor_sum(a, 32);
        b = wave_xor_sum(b, 16); b = wave_xor_sum(b, 32);
        if (lq == 0) scr[((s2_pix0 + i) * 16 + lrow) * (NW / 2) + s2_cg] = make_float2(a, b);
      }
      tb_barrier();
      if (lane < 16) {
        const int rl = 16 * wave + lane;                 // row of the tile: block rl / 16, row rl % 16 (NW == 8: 128 rows)
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int cgi = 0; cgi < NW / 2; ++cgi) {
          const float2 t = scr[rl * (NW / 2) + cgi];
          a += t.x;
          b += t.y;
        }
        const float mean = a * (1.0f / TB_C);
        const float var = fmaxf(b * (1.0f / TB_C) - mean * mean, 0.f);
        const int fr = HB == 1 ? (rl & 15) : 16 * ((rl >> 4) & 1) + (rl & 15);
        if (fr < F)
          reinterpret_cast<float2*>(p.stats_out)[row0 + blk_pix(rl >> 4) + (size_t)fr * p.hw] = make_float2(mean, rsqrtf(var + p.ln_eps));
      }
    }
  }
  tb_wait_vm<0>();   // run-ahead copies must land before the LDS is handed on
}

// ---- pack: folded [3 C, C] QKV weight -> fragment-major chunks; [C, C] out-projection -> the three parts per head pair;
// bias + positional rows -> the 16 x 16 table behind each chunk's weights; column sums -> packed order
__global__ void tblock_pack_kernel(const bf16_t* __restrict__ wqkv, const float* __restrict__ bias,
                                   const float* __restrict__ colsum, const float* __restrict__ pe, int pe_ld,
                                   const bf16_t* __restrict__ wo, char* __restrict__ wqkv_t, bf16_t* __restrict__ wo_t,
                                   float* __restrict__ colsum_p, int f) {
  const int HB = f > 16 ? 2 : 1;                                 // table halves per head (TbGeo<F>::HB)
  const int TB_SLOT = TB_WBYTES + 2 * HB * 1024;
  const int n_w = TB_QKV_CHUNKS * (TB_WBYTES / 16);              // 16-byte weight items of the QKV stream
  const int n_t = TB_QKV_CHUNKS * 2 * HB * 256;                  // table floats of the QKV stream
  const int n_o01 = 4 * 2 * 20 * 64, n_o2 = 4 * 20 * 64;         // 16-byte items of parts 0 / 1, 8-byte items of part 2
  const int total = n_w + n_t + n_o01 + n_o2 + TB_PCOLS;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    if (idx < n_w) {
      // chunk = 8 hp + blk: [ks][head of the pair][lane][8]
      const int lane = idx & 63, wn = (idx >> 6) & 1, ks = (idx >> 7) % TB_KS, ch = idx / (128 * TB_KS);
      const int hp = ch >> 3, blk = ch & 7, head = 2 * hp + wn;
      const int col = tb_src_col(head, blk, lane & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (col >= 0) v = *reinterpret_cast<const uint4*>(wqkv + (size_t)col * TB_C + 32 * ks + 8 * (lane >> 4));
      *reinterpret_cast<uint4*>(wqkv_t + (size_t)ch * TB_SLOT + (size_t)(idx - ch * (128 * TB_KS)) * 16) = v;
    } else if (idx < n_w + n_t) {
      // behind the chunk's weights: [head of the pair][block half][16][16] fp32 = bias + positional row of frame 16 half +
      // row, [row][column] for the transposed blocks (0..4), [column][row] for the V blocks; frames >= f (padding): 0
      const int k = idx - n_w;
      const int e = k & 255, hh = (k >> 8) % HB, wn = ((k >> 8) / HB) & 1, ch = k / (512 * HB);
      const int hp = ch >> 3, blk = ch & 7, head = 2 * hp + wn;
      const int fr = 16 * hh + (blk < 5 ? e >> 4 : e & 15), r = blk < 5 ? e & 15 : e >> 4;
      const int col = tb_src_col(head, blk, r);
      float v = 0.f;
      if (col >= 0 && fr < f) v = (bias != nullptr ? bias[col] : 0.f) + (pe != nullptr ? pe[(size_t)fr * pe_ld + col] : 0.f);
      *reinterpret_cast<float*>(wqkv_t + (size_t)ch * TB_SLOT + TB_WBYTES + (wn * HB + hh) * 1024 + e * 4) = v;
    } else if (idx < n_w + n_t + n_o01) {
      // [hp][part 0 / 1][column block j][lane][8]: out column 16 j + (lane & 15); k slots 8 lq .. + 7 =
      // channels 4 lq .. + 3 and 16 + 4 lq .. + 3 of head 2 hp + part
      const int k = idx - n_w - n_t;
      const int lane = k & 63, j = (k >> 6) % 20, part = (k / (64 * 20)) & 1, hp = k / (64 * 20 * 2);
      const int lq = lane >> 4, head = 2 * hp + part;
      const bf16_t* src = wo + (size_t)(16 * j + (lane & 15)) * TB_C + head * TB_D;
      bf16_t* dst = wo_t + (size_t)hp * (WO_PAIR / 2) + (size_t)part * (WO_P01 / 2) + (size_t)(j * 64 + lane) * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dst[e] = src[4 * lq + e];
        dst[4 + e] = src[16 + 4 * lq + e];
      }
    } else if (idx < n_w + n_t + n_o01 + n_o2) {
      // [hp][part 2][column block j][lane][4]: k slots 4 lq .. + 3 = channels 32 + 4 lq .. of head 2 hp (lq < 2),
      // 32 + 4 (lq - 2) .. of head 2 hp + 1 (lq >= 2)
      const int k = idx - n_w - n_t - n_o01;
      const int lane = k & 63, j = (k >> 6) % 20, hp = k / (64 * 20);
      const int lq = lane >> 4, head = 2 * hp + (lq >> 1);
      const bf16_t* src = wo + (size_t)(16 * j + (lane & 15)) * TB_C + head * TB_D + 32 + 4 * (lq & 1);
      bf16_t* dst = wo_t + (size_t)hp * (WO_PAIR / 2) + (size_t)(2 * WO_P01 / 2) + (size_t)(j * 64 + lane) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = src[e];
    } else {
      const int pc = idx - (n_w + n_t + n_o01 + n_o2);
      const int col = tb_src_col(pc >> 7, (pc >> 4) & 7, pc & 15);
      colsum_p[pc] = col >= 0 ? colsum[col] : 0.f;
    }
  }
}

}  // namespace

extern "C" int vx_tblock_pack(const void* wqkv, const float* bias, const float* colsum, const float* pe_rows, int pe_ld,
                              const void* wo, void* wqkv_t, void* wo_t, float* colsum_p, int c, int heads, int f,
                              void* stream_) {
  VX_REQUIRE(wqkv != nullptr && colsum != nullptr && wo != nullptr && wqkv_t != nullptr && wo_t != nullptr &&
                 colsum_p != nullptr, "vx_tblock_pack: null pointer");
  VX_REQUIRE(c == TB_C && heads == TB_HEADS && (f == 16 || f == 24),
             "vx_tblock_pack: only C = %d, %d heads, 16 or 24 frames (the 64x64 level) is built", TB_C, TB_HEADS);
  VX_REQUIRE(pe_rows == nullptr || pe_ld >= 3 * TB_C, "vx_tblock_pack: pe_ld=%d", pe_ld);
  hipLaunchKernelGGL(tblock_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)wqkv, bias, colsum,
                     pe_rows, pe_ld, (const bf16_t*)wo, (char*)wqkv_t, (bf16_t*)wo_t, colsum_p, f);
  return vx_check_launch("vx_tblock_pack");
}

extern "C" int64_t vx_tblock_packed_bytes(int f) { return (int64_t)TB_QKV_CHUNKS * (TB_WBYTES + 2 * (f > 16 ? 2 : 1) * 1024); }

template <int F>
static int tblock_launch(const vx_tblock_params& p, hipStream_t stream) {
  using G_ = TbGeo<F>;
  static bool attr_set = false;
  static int cus = 256;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tblock_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS);
    if (e != hipSuccess) {
      vx_set_error("vx_tblock_fused: hipFuncSetAttribute(%d B LDS) failed: %s", G_::LDS, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    attr_set = true;
  }
  const int tiles = p.b * (p.hw / G_::PIX);
  g_vx_last_kernel = F == 16 ? "tblock_kernel<16>" : "tblock_kernel<24>";
  hipLaunchKernelGGL(tblock_kernel<F>, dim3(tiles < cus ? tiles : cus), dim3(64 * TB_NW), G_::LDS, stream, p,
                     p.scale * 1.4426950408889634f);
  return vx_check_launch("vx_tblock_fused");
}

extern "C" int vx_tblock_fused(const vx_tblock_params* pp, void* stream_) {
  const vx_tblock_params& p = *pp;
  VX_REQUIRE(p.x != nullptr && p.wqkv_t != nullptr && p.wo_t != nullptr && p.colsum_p != nullptr,
             "vx_tblock_fused: null pointer");
  VX_REQUIRE(p.c == TB_C && p.heads == TB_HEADS && (p.f == 16 || p.f == 24),
             "vx_tblock_fused: only C = %d, %d heads, 16 or 24 frames (the 64x64 level) is built", TB_C, TB_HEADS);
  const int pix = p.f == 16 ? TbGeo<16>::PIX : TbGeo<24>::PIX;
  VX_REQUIRE(p.b > 0 && p.hw > 0 && (p.hw % pix) == 0, "vx_tblock_fused: hw=%d must be a multiple of %d", p.hw, pix);
  VX_REQUIRE((p.ldx % 8) == 0 && p.ldx >= TB_C, "vx_tblock_fused: row stride");
  VX_REQUIRE((unsigned long long)p.b * p.f * p.hw * p.ldx < (1ull << 32), "vx_tblock_fused: more than 4 G elements");
  return p.f == 16 ? tblock_launch<16>(p, (hipStream_t)stream_) : tblock_launch<24>(p, (hipStream_t)stream_);
}
