// Audio cross-attention of a spatial transformer block as ONE streaming launch for gfx950 (round 6):
//     h <- h + alpha * ( to_out( softmax_t( q k_t^T / sqrt d ) v ) ),   q = to_q(LayerNorm(h)),  k, v = to_k / to_v(audio tokens)
// (modules/mutual_self_attention.py:227-244 -> diffusers Attention / AttnProcessor2_0 with encoder_hidden_states = the
// frame's audio tokens).  The keys of a frame are FIVE tokens (AudioProjection num_queries), so the whole block is two skinny
// products whose "weights" depend on the frame only and are built once per clip (vx_audio_xattn_pack):
//     S[m, (h, t)]   = rstd_m * ( sum_c x[m, c] Kq_f[(h, t), c] - mean_m * colsum_f[(h, t)] ) + sbias_f[(h, t)]
//         Kq_f[(h, t), c] = log2(e) / sqrt(d) * sum_j K_f[t, h d + j] * wq[h d + j, c]        (wq: LayerNorm-folded to_q weight)
//     P = softmax over the 5 tokens of every head (base 2: the log2(e) is in Kq)
//     y[m, n] = x[m, n] + alpha * ( sum_(h,t) P[m, (h, t)] VO_f[n, (h, t)] + bias_o[n] ),  VO_f[n, (h, t)] = sum_j wo[n, h d + j] V_f[t, h d + j]
// i.e. 2 x 48 columns per row instead of the two C x C projections (q, to_out) around a 5-key attention: the [rows, C] query
// and attention-output tensors never exist, the FLOPs drop by C / 48, and the launch reads x once and writes y once
// (HBM-bound; three launches of 30-40 us each at every level before).
//
// Column ("slot") order of the 8 heads x 5 tokens, chosen so that the softmax is lane-local in the MFMA accumulator layout
// (lane = row l & 15, columns 16 j + 4 (l >> 4) + r):  slots 0-15: head (l >> 4), token r;  16-31: head 4 + (l >> 4), token r;
// 32-47: r = 0 token 4 of head (l >> 4), r = 1 token 4 of head 4 + (l >> 4), r = 2, 3 padding (zero weights, P = 0).
// The probabilities in that layout ARE the K = 16 operand of v_mfma_f32_16x16x16: no lane exchange, no LDS, no barrier.
//
// One wave = MI 16-row blocks of one frame; x fragments, Kq fragments (fragment-major packed, 1 KiB per (k-step, 16 slots))
// and VO fragments (512 B per (16 output columns, 16 slots)) come straight from global memory / L2 with PF loads in flight.
#include "vx_common.h"
#include "../../include/vexpress_hip.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int AX_HEADS = 8, AX_TOK = 5, AX_SLOTS = 48;
constexpr int AX_PF1 = 5;      // k-steps of stage 1 in flight (C / 32 = 10, 20, 40 k-steps: multiples of 5)
constexpr int AX_PF2 = 4;      // 16-column output blocks of stage 2 in flight (C / 16 = 20, 40, 80)

// slot of (head, token) - see the head comment
__host__ __device__ inline int ax_slot(int h, int t) { return t < 4 ? 16 * (h >> 2) + 4 * (h & 3) + t : 32 + 4 * (h & 3) + (h >> 2); }

template <int MI>
__global__ __launch_bounds__(256) void axattn_kernel(const vx_axattn_params p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int row0 = ((int)blockIdx.x * 4 + wave) * (16 * MI);
  if (row0 >= p.rows) return;                       // (no barriers in this kernel: waves leave one by one)
  const int C = p.c, KS = C >> 5, NB = C >> 4;
  const int frame = row0 / p.rows_per_frame;
  const bf16_t* x = (const bf16_t*)p.x;             // (may alias out: no __restrict__)
  bf16_t* out = (bf16_t*)p.out;

  // ---- LayerNorm statistics of the wave's rows (vx_gemm_params.ln_stats formats)
  float rs[MI], rm[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = row0 + 16 * i + lrow;
    if (p.ln_stats_parts == 2) {
      const float4 t = reinterpret_cast<const float4*>(p.ln_stats)[row];
      const float inv = 1.0f / (float)C;
      const float mean = (t.x + t.z) * inv;
      float var = (t.y + t.w) * inv - mean * mean;
      var = var > 0.f ? var : 0.f;
      rs[i] = 1.0f / sqrtf(var + p.ln_eps);
      rm[i] = -mean * rs[i];
    } else {
      const float2 t = reinterpret_cast<const float2*>(p.ln_stats)[row];
      rs[i] = t.y;
      rm[i] = -t.x * t.y;
    }
  }

  // ---- stage 1: S = x Kq^T over C, AX_PF1 k-steps of loads in flight
  const bf16_t* xr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) xr[i] = x + (size_t)(row0 + 16 * i + lrow) * p.ldx + 8 * lq;
  const bf16_t* kqf = (const bf16_t*)p.kq + (size_t)frame * KS * (3 * 512) + lane * 8;
  f32x4_t S[MI][3];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 xa[AX_PF1][MI], kf[AX_PF1][3];
  auto issue1 = [&](int ks, int d) {
#pragma unroll
    for (int i = 0; i < MI; ++i) xa[d][i] = *reinterpret_cast<const uint4*>(xr[i] + 32 * ks);
#pragma unroll
    for (int j = 0; j < 3; ++j) kf[d][j] = *reinterpret_cast<const uint4*>(kqf + (size_t)(ks * 3 + j) * 512);
  };
#pragma unroll
  for (int d = 0; d < AX_PF1; ++d) issue1(d, d);
  for (int k0 = 0; k0 < KS; k0 += AX_PF1) {
#pragma unroll
    for (int d = 0; d < AX_PF1; ++d) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) S[i][j] = mfma16(kf[d][j], xa[d][i], S[i][j]);
      if (k0 + d + AX_PF1 < KS) issue1(k0 + d + AX_PF1, d);
    }
  }

  // ---- LayerNorm fold + per-head softmax over 5 tokens, lane-local; P as the K = 16 MFMA operand
  const float* csf = p.kq_colsum + (size_t)frame * AX_SLOTS + 4 * lq;
  const float* sbf = p.kq_bias + (size_t)frame * AX_SLOTS + 4 * lq;
  vx_e16x4_t P[MI][3];
  {
    float4 cs[3], sb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      cs[j] = *reinterpret_cast<const float4*>(csf + 16 * j);
      sb[j] = *reinterpret_cast<const float4*>(sbf + 16 * j);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float a[5], b[5];     // head lq / head 4 + lq
      a[0] = fmaf(rs[i], S[i][0][0], fmaf(rm[i], cs[0].x, sb[0].x));
      a[1] = fmaf(rs[i], S[i][0][1], fmaf(rm[i], cs[0].y, sb[0].y));
      a[2] = fmaf(rs[i], S[i][0][2], fmaf(rm[i], cs[0].z, sb[0].z));
      a[3] = fmaf(rs[i], S[i][0][3], fmaf(rm[i], cs[0].w, sb[0].w));
      a[4] = fmaf(rs[i], S[i][2][0], fmaf(rm[i], cs[2].x, sb[2].x));
      b[0] = fmaf(rs[i], S[i][1][0], fmaf(rm[i], cs[1].x, sb[1].x));
      b[1] = fmaf(rs[i], S[i][1][1], fmaf(rm[i], cs[1].y, sb[1].y));
      b[2] = fmaf(rs[i], S[i][1][2], fmaf(rm[i], cs[1].z, sb[1].z));
      b[3] = fmaf(rs[i], S[i][1][3], fmaf(rm[i], cs[1].w, sb[1].w));
      b[4] = fmaf(rs[i], S[i][2][1], fmaf(rm[i], cs[2].y, sb[2].y));
      const float ma = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), a[4]);
      const float mb = fmaxf(fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])), b[4]);
      float la = 0.f, lb = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        a[t] = __builtin_amdgcn_exp2f(a[t] - ma);
        b[t] = __builtin_amdgcn_exp2f(b[t] - mb);
        la += a[t];
        lb += b[t];
      }
      const float ia = __builtin_amdgcn_rcpf(la), ib = __builtin_amdgcn_rcpf(lb);
      const uint2 pa = make_uint2(pack_bf16x2(a[0] * ia, a[1] * ia), pack_bf16x2(a[2] * ia, a[3] * ia));
      const uint2 pb = make_uint2(pack_bf16x2(b[0] * ib, b[1] * ib), pack_bf16x2(b[2] * ib, b[3] * ib));
      const uint2 pc = make_uint2(pack_bf16x2(a[4] * ia, b[4] * ib), 0u);
      P[i][0] = __builtin_bit_cast(vx_e16x4_t, pa);
      P[i][1] = __builtin_bit_cast(vx_e16x4_t, pb);
      P[i][2] = __builtin_bit_cast(vx_e16x4_t, pc);
    }
  }

  // ---- stage 2: y = x + alpha (P VO^T + bias_o), one 16-column block at a time, AX_PF2 blocks of loads in flight
  const bf16_t* vof = (const bf16_t*)p.vo + (size_t)frame * NB * (3 * 256) + lane * 4;
  const bf16_t* rr[MI];
  bf16_t* orow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    rr[i] = x + (size_t)(row0 + 16 * i + lrow) * p.ldx + 4 * lq;
    orow[i] = out + (size_t)(row0 + 16 * i + lrow) * p.ldo + 4 * lq;
  }
  const float* bo = p.bias_o + 4 * lq;
  uint2 wv[AX_PF2][3], rv[AX_PF2][MI];
  float4 bv[AX_PF2];
  auto issue2 = [&](int nb, int d) {
#pragma unroll
    for (int jk = 0; jk < 3; ++jk) wv[d][jk] = *reinterpret_cast<const uint2*>(vof + (size_t)(nb * 3 + jk) * 256);
#pragma unroll
    for (int i = 0; i < MI; ++i) rv[d][i] = *reinterpret_cast<const uint2*>(rr[i] + 16 * nb);
    bv[d] = *reinterpret_cast<const float4*>(bo + 16 * nb);
  };
  const float alpha = p.alpha;
  const bool want_stats = p.row_stats_out != nullptr;
  const int half_nb = NB >> 1;
  float s0[MI], q0[MI], s1[MI], q1[MI];      // (sum, sum of squares) of the stored values: first / second half of the row
#pragma unroll
  for (int i = 0; i < MI; ++i) s0[i] = q0[i] = s1[i] = q1[i] = 0.f;
#pragma unroll
  for (int d = 0; d < AX_PF2; ++d) issue2(d, d);
  for (int n0 = 0; n0 < NB; n0 += AX_PF2) {
#pragma unroll
    for (int d = 0; d < AX_PF2; ++d) {
      const int nb = n0 + d;
      const bool first_half = nb < half_nb;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jk = 0; jk < 3; ++jk)
          acc = VX_MFMA_16x16x16(__builtin_bit_cast(vx_e16x4_t, wv[d][jk]), P[i][jk], acc, 0, 0, 0);
        const float y0 = fmaf(alpha, acc[0] + bv[d].x, e16_lo(rv[d][i].x));
        const float y1 = fmaf(alpha, acc[1] + bv[d].y, e16_hi(rv[d][i].x));
        const float y2 = fmaf(alpha, acc[2] + bv[d].z, e16_lo(rv[d][i].y));
        const float y3 = fmaf(alpha, acc[3] + bv[d].w, e16_hi(rv[d][i].y));
        const uint2 pk = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
        *reinterpret_cast<uint2*>(orow[i] + 16 * nb) = pk;
        if (want_stats) {
          // statistics of the STORED (rounded) values, as vx_row_stats would read them back
          const float r0 = e16_lo(pk.x), r1 = e16_hi(pk.x), r2 = e16_lo(pk.y), r3 = e16_hi(pk.y);
          const float s = (r0 + r1) + (r2 + r3);
          float q = r0 * r0;
          q = fmaf(r1, r1, q); q = fmaf(r2, r2, q); q = fmaf(r3, r3, q);
          if (first_half) { s0[i] += s; q0[i] += q; } else { s1[i] += s; q1[i] += q; }
        }
      }
      if (nb + AX_PF2 < NB) issue2(nb + AX_PF2, d);
    }
  }
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // the four lanes {l, l ^ 16, l ^ 32, l ^ 48} share a row: fixed order -> the same bits for any batch
      float a0 = s0[i], b0 = q0[i], a1 = s1[i], b1 = q1[i];
      a0 += __shfl_xor(a0, 16, 64); b0 += __shfl_xor(b0, 16, 64); a1 += __shfl_xor(a1, 16, 64); b1 += __shfl_xor(b1, 16, 64);
      a0 += __shfl_xor(a0, 32, 64); b0 += __shfl_xor(b0, 32, 64); a1 += __shfl_xor(a1, 32, 64); b1 += __shfl_xor(b1, 32, 64);
      if (lq == 0) {
        const int row = row0 + 16 * i + lrow;
        if (p.row_stats_parts == 2) {
          reinterpret_cast<float4*>(p.row_stats_out)[row] = make_float4(a0, b0, a1, b1);
        } else {
          const float inv = 1.0f / (float)C;
          const float mean = (a0 + a1) * inv;
          float var = (b0 + b1) * inv - mean * mean;
          var = var > 0.f ? var : 0.f;
          reinterpret_cast<float2*>(p.row_stats_out)[row] = make_float2(mean, 1.0f / sqrtf(var + p.row_stats_eps));
        }
      }
    }
  }
}

// The same block for launches with few rows (the 32x32 ... 8x8 levels: 16384 ... 1024 rows), where one wave per 16 rows
// leaves most SIMDs empty and every wave walks 40 + 80 dependent load groups at C = 1280: the FOUR waves of a workgroup share
// one 16-row block - stage 1 split over the k-steps (wave w takes k-steps w, w + 4, ...; the partial S meet in LDS and are
// added in wave order by everyone: same bits in all four), the softmax computed by each wave, stage 2 split over the output
// columns (wave w owns the w-th quarter of the row: for the two-part statistics waves 0, 1 are the first half).
constexpr int AXS_PF = 5;
__global__ __launch_bounds__(256) void axattn_split_kernel(const vx_axattn_params p) {
  __shared__ float4 s_part[4][3][64];
  __shared__ float2 s_stat[4][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int row0 = (int)blockIdx.x * 16;
  const int C = p.c, KS = C >> 5, NB = C >> 4;
  const int frame = row0 / p.rows_per_frame;
  const bf16_t* x = (const bf16_t*)p.x;
  bf16_t* out = (bf16_t*)p.out;
  const int row = row0 + lrow;

  float rs, rm;
  if (p.ln_stats_parts == 2) {
    const float4 t = reinterpret_cast<const float4*>(p.ln_stats)[row];
    const float inv = 1.0f / (float)C;
    const float mean = (t.x + t.z) * inv;
    float var = (t.y + t.w) * inv - mean * mean;
    var = var > 0.f ? var : 0.f;
    rs = 1.0f / sqrtf(var + p.ln_eps);
    rm = -mean * rs;
  } else {
    const float2 t = reinterpret_cast<const float2*>(p.ln_stats)[row];
    rs = t.y;
    rm = -t.x * t.y;
  }

  // ---- stage 1: this wave's k-steps wave, wave + 4, ...
  const bf16_t* xr = x + (size_t)row * p.ldx + 8 * lq;
  const bf16_t* kqf = (const bf16_t*)p.kq + (size_t)frame * KS * (3 * 512) + lane * 8;
  f32x4_t S[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) S[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int my_ks = (KS - wave + 3) >> 2;
  for (int g0 = 0; g0 < my_ks; g0 += AXS_PF) {
    uint4 xa[AXS_PF], kf[AXS_PF][3];
#pragma unroll
    for (int d = 0; d < AXS_PF; ++d) {
      if (g0 + d < my_ks) {
        const int ks = wave + 4 * (g0 + d);
        xa[d] = *reinterpret_cast<const uint4*>(xr + 32 * ks);
#pragma unroll
        for (int j = 0; j < 3; ++j) kf[d][j] = *reinterpret_cast<const uint4*>(kqf + (size_t)(ks * 3 + j) * 512);
      }
    }
#pragma unroll
    for (int d = 0; d < AXS_PF; ++d) {
      if (g0 + d < my_ks) {
#pragma unroll
        for (int j = 0; j < 3; ++j) S[j] = mfma16(kf[d][j], xa[d], S[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) s_part[wave][j][lane] = make_float4(S[j][0], S[j][1], S[j][2], S[j][3]);
  __syncthreads();
  float sv[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 a0 = s_part[0][j][lane], a1 = s_part[1][j][lane], a2 = s_part[2][j][lane], a3 = s_part[3][j][lane];
    sv[j][0] = ((a0.x + a1.x) + a2.x) + a3.x;
    sv[j][1] = ((a0.y + a1.y) + a2.y) + a3.y;
    sv[j][2] = ((a0.z + a1.z) + a2.z) + a3.z;
    sv[j][3] = ((a0.w + a1.w) + a2.w) + a3.w;
  }

  // ---- LayerNorm fold + softmax (every wave: same inputs, same bits)
  vx_e16x4_t P[3];
  {
    const float* csf = p.kq_colsum + (size_t)frame * AX_SLOTS + 4 * lq;
    const float* sbf = p.kq_bias + (size_t)frame * AX_SLOTS + 4 * lq;
    float4 cs[3], sb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      cs[j] = *reinterpret_cast<const float4*>(csf + 16 * j);
      sb[j] = *reinterpret_cast<const float4*>(sbf + 16 * j);
    }
    float a[5], b[5];
    a[0] = fmaf(rs, sv[0][0], fmaf(rm, cs[0].x, sb[0].x));
    a[1] = fmaf(rs, sv[0][1], fmaf(rm, cs[0].y, sb[0].y));
    a[2] = fmaf(rs, sv[0][2], fmaf(rm, cs[0].z, sb[0].z));
    a[3] = fmaf(rs, sv[0][3], fmaf(rm, cs[0].w, sb[0].w));
    a[4] = fmaf(rs, sv[2][0], fmaf(rm, cs[2].x, sb[2].x));
    b[0] = fmaf(rs, sv[1][0], fmaf(rm, cs[1].x, sb[1].x));
    b[1] = fmaf(rs, sv[1][1], fmaf(rm, cs[1].y, sb[1].y));
    b[2] = fmaf(rs, sv[1][2], fmaf(rm, cs[1].z, sb[1].z));
    b[3] = fmaf(rs, sv[1][3], fmaf(rm, cs[1].w, sb[1].w));
    b[4] = fmaf(rs, sv[2][1], fmaf(rm, cs[2].y, sb[2].y));
    const float ma = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), a[4]);
    const float mb = fmaxf(fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])), b[4]);
    float la = 0.f, lb = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      a[t] = __builtin_amdgcn_exp2f(a[t] - ma);
      b[t] = __builtin_amdgcn_exp2f(b[t] - mb);
      la += a[t];
      lb += b[t];
    }
    const float ia = __builtin_amdgcn_rcpf(la), ib = __builtin_amdgcn_rcpf(lb);
    P[0] = __builtin_bit_cast(vx_e16x4_t, make_uint2(pack_bf16x2(a[0] * ia, a[1] * ia), pack_bf16x2(a[2] * ia, a[3] * ia)));
    P[1] = __builtin_bit_cast(vx_e16x4_t, make_uint2(pack_bf16x2(b[0] * ib, b[1] * ib), pack_bf16x2(b[2] * ib, b[3] * ib)));
    P[2] = __builtin_bit_cast(vx_e16x4_t, make_uint2(pack_bf16x2(a[4] * ia, b[4] * ib), 0u));
  }

  // ---- stage 2: this wave's quarter of the output columns
  const int nbq = NB >> 2, nb_lo = wave * nbq;
  const bf16_t* vof = (const bf16_t*)p.vo + (size_t)frame * NB * (3 * 256) + lane * 4;
  const bf16_t* rr = x + (size_t)row * p.ldx + 4 * lq;
  bf16_t* orow = out + (size_t)row * p.ldo + 4 * lq;
  const float* bo = p.bias_o + 4 * lq;
  const float alpha = p.alpha;
  float ssum = 0.f, ssq = 0.f;
  for (int n0 = nb_lo; n0 < nb_lo + nbq; n0 += AXS_PF) {
    uint2 wv[AXS_PF][3], rv[AXS_PF];
    float4 bv[AXS_PF];
#pragma unroll
    for (int d = 0; d < AXS_PF; ++d) {
      const int nb = n0 + d;
#pragma unroll
      for (int jk = 0; jk < 3; ++jk) wv[d][jk] = *reinterpret_cast<const uint2*>(vof + (size_t)(nb * 3 + jk) * 256);
      rv[d] = *reinterpret_cast<const uint2*>(rr + 16 * nb);
      bv[d] = *reinterpret_cast<const float4*>(bo + 16 * nb);
    }
#pragma unroll
    for (int d = 0; d < AXS_PF; ++d) {
      const int nb = n0 + d;
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jk = 0; jk < 3; ++jk) acc = VX_MFMA_16x16x16(__builtin_bit_cast(vx_e16x4_t, wv[d][jk]), P[jk], acc, 0, 0, 0);
      const float y0 = fmaf(alpha, acc[0] + bv[d].x, e16_lo(rv[d].x));
      const float y1 = fmaf(alpha, acc[1] + bv[d].y, e16_hi(rv[d].x));
      const float y2 = fmaf(alpha, acc[2] + bv[d].z, e16_lo(rv[d].y));
      const float y3 = fmaf(alpha, acc[3] + bv[d].w, e16_hi(rv[d].y));
      const uint2 pk = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
      *reinterpret_cast<uint2*>(orow + 16 * nb) = pk;
      const float r0 = e16_lo(pk.x), r1 = e16_hi(pk.x), r2 = e16_lo(pk.y), r3 = e16_hi(pk.y);
      ssum += (r0 + r1) + (r2 + r3);
      float q = r0 * r0;
      q = fmaf(r1, r1, q); q = fmaf(r2, r2, q); q = fmaf(r3, r3, q);
      ssq += q;
    }
  }
  if (p.row_stats_out != nullptr) {
    ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
    ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
    if (lq == 0) s_stat[wave][lrow] = make_float2(ssum, ssq);
    __syncthreads();
    if (tid < 16) {
      const float2 t0 = s_stat[0][tid], t1 = s_stat[1][tid], t2 = s_stat[2][tid], t3 = s_stat[3][tid];
      const float a0 = t0.x + t1.x, b0 = t0.y + t1.y, a1 = t2.x + t3.x, b1 = t2.y + t3.y;
      if (p.row_stats_parts == 2) {
        reinterpret_cast<float4*>(p.row_stats_out)[row0 + tid] = make_float4(a0, b0, a1, b1);
      } else {
        const float inv = 1.0f / (float)C;
        const float mean = (a0 + a1) * inv;
        float var = (b0 + b1) * inv - mean * mean;
        var = var > 0.f ? var : 0.f;
        reinterpret_cast<float2*>(p.row_stats_out)[row0 + tid] = make_float2(mean, 1.0f / sqrtf(var + p.row_stats_eps));
      }
    }
  }
}

// ---- once per clip: the per-frame operands.  Both products are [frames * 5 tokens, d] x [d, C] per head - tiny, but a
// block per output row re-reads the head's weight slice 80 times (first version: 98 + 191 us per transformer block, 4.6 ms per
// clip).  Here a block = (head, 64 output columns / rows, 16 frames): the head's K (or V) rows of those frames sit in LDS as
// float (80 x d <= 51 KB), the weight slice is read ONCE, and every thread owns one column x 20 (frame, token) rows.
constexpr int AXP_FR = 16;                       // frames per block
constexpr int AXP_ROWS = AXP_FR * AX_TOK;        // 80 (frame, token) rows
constexpr int AXP_RPT = AXP_ROWS / 4;            // 20 rows per thread (4 row groups x 64 columns = 256 threads)

__device__ __forceinline__ int ax_kq_index(int KS, int frame, int slot, int col) {
  const int ks = col >> 5, fgrp = (col & 31) >> 3, e = col & 7;
  return ((frame * KS + ks) * 3 + (slot >> 4)) * 512 + (fgrp * 16 + (slot & 15)) * 8 + e;
}

// grid (8 heads, C / 64, ceil(frames / 16)); dynamic LDS 80 * d floats
__global__ __launch_bounds__(256) void axattn_pack_kq_kernel(const bf16_t* __restrict__ kv, int ldkv, const bf16_t* __restrict__ wq,
                                                             int c, int d, float scale, int frames, bf16_t* __restrict__ kq) {
  extern __shared__ float kvals[];               // [80][d]
  const int h = blockIdx.x, col = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, f0 = blockIdx.z * AXP_FR;
  const int nrows = min(AXP_FR, frames - f0) * AX_TOK;
  for (int i = threadIdx.x; i < AXP_ROWS * d; i += 256) {
    const int r = i / d, j = i - r * d;
    kvals[i] = r < nrows ? bf16_to_f32(kv[(size_t)(f0 * AX_TOK + r) * ldkv + h * d + j]) : 0.f;
  }
  __syncthreads();
  float acc[AXP_RPT];
#pragma unroll
  for (int r = 0; r < AXP_RPT; ++r) acc[r] = 0.f;
  const bf16_t* wcol = wq + (size_t)(h * d) * c + col;
  const float* kr = kvals + (size_t)(rg * AXP_RPT) * d;
#pragma unroll 4
  for (int j = 0; j < d; ++j) {
    const float w = bf16_to_f32(wcol[(size_t)j * c]);
#pragma unroll
    for (int r = 0; r < AXP_RPT; ++r) acc[r] = fmaf(kr[r * d + j], w, acc[r]);
  }
  const int KS = c >> 5;
#pragma unroll
  for (int r = 0; r < AXP_RPT; ++r) {
    const int row = rg * AXP_RPT + r;
    if (row < nrows) {
      const int fr = f0 + row / AX_TOK, t = row % AX_TOK;
      kq[ax_kq_index(KS, fr, ax_slot(h, t), col)] = f32_to_bf16(acc[r] * scale);
    }
  }
}

// grid (frames, 48): the padding slots' zero rows, the column sums of the ROUNDED values the MFMA sees (like
// weights.fold_layernorm; fixed-order tree) and the bias  sbias[(h, t)] = scale * sum_j K[t, h d + j] bq[h d + j]
__global__ __launch_bounds__(256) void axattn_pack_sums_kernel(const bf16_t* __restrict__ kv, int ldkv, const float* __restrict__ bq,
                                                               int c, int d, float scale, bf16_t* __restrict__ kq,
                                                               float* __restrict__ colsum, float* __restrict__ sbias) {
  __shared__ float red[256];
  const int frame = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x;
  const int KS = c >> 5;
  int h = -1, t = -1;            // inverse of ax_slot
  if (slot < 32) {
    h = 4 * (slot >> 4) + ((slot & 15) >> 2);
    t = slot & 3;
  } else if (((slot - 32) & 3) < 2) {
    h = ((slot - 32) >> 2) + 4 * ((slot - 32) & 3);
    t = 4;
  }
  float part = 0.f;
  for (int col = tid; col < c; col += 256) {
    const int idx = ax_kq_index(KS, frame, slot, col);
    if (h < 0) kq[idx] = 0;
    else part += bf16_to_f32(kq[idx]);
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float cs = red[0];
  __syncthreads();
  float sb = 0.f;
  if (h >= 0 && bq != nullptr)
    for (int j = tid; j < d; j += 256) sb = fmaf(bf16_to_f32(kv[(size_t)(frame * AX_TOK + t) * ldkv + h * d + j]), bq[h * d + j], sb);
  red[tid] = sb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    colsum[(size_t)frame * AX_SLOTS + slot] = cs;
    sbias[(size_t)frame * AX_SLOTS + slot] = red[0] * scale;
  }
}

// grid (8 heads, C / 64 output channels, ceil(frames / 16)); thread = one output channel n x 20 (frame, token) rows; the
// thread's weight row segment wo[n, h d .. h d + d) is read once, 16 bytes at a time
__global__ __launch_bounds__(256) void axattn_pack_vo_kernel(const bf16_t* __restrict__ kv, int ldkv, const bf16_t* __restrict__ wo,
                                                             int c, int d, int frames, bf16_t* __restrict__ vo) {
  extern __shared__ float vvals[];               // [80][d]
  const int h = blockIdx.x, n = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, f0 = blockIdx.z * AXP_FR;
  const int nrows = min(AXP_FR, frames - f0) * AX_TOK;
  for (int i = threadIdx.x; i < AXP_ROWS * d; i += 256) {
    const int r = i / d, j = i - r * d;
    vvals[i] = r < nrows ? bf16_to_f32(kv[(size_t)(f0 * AX_TOK + r) * ldkv + c + h * d + j]) : 0.f;
  }
  __syncthreads();
  float acc[AXP_RPT];
#pragma unroll
  for (int r = 0; r < AXP_RPT; ++r) acc[r] = 0.f;
  const bf16_t* wrow = wo + (size_t)n * c + h * d;          // d % 8 == 0 (c % 320 == 0, 8 heads), rows 16-byte aligned
  const float* vr = vvals + (size_t)(rg * AXP_RPT) * d;
  for (int j8 = 0; j8 < d; j8 += 8) {
    float w[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(wrow + j8), w);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int r = 0; r < AXP_RPT; ++r) acc[r] = fmaf(vr[r * d + j8 + e], w[e], acc[r]);
  }
  const int NB = c >> 4, nb = n >> 4, frow = n & 15;
#pragma unroll
  for (int r = 0; r < AXP_RPT; ++r) {
    const int row = rg * AXP_RPT + r;
    if (row < nrows) {
      const int fr = f0 + row / AX_TOK, t = row % AX_TOK;
      const int slot = ax_slot(h, t);
      vo[((size_t)(fr * NB + nb) * 3 + (slot >> 4)) * 256 + (((slot & 15) >> 2) * 16 + frow) * 4 + (slot & 3)] = f32_to_bf16(acc[r]);
    }
  }
  // the padding slots 32 + 4 q + 2, + 3 (q = 0 .. 3) of this thread's row: heads 0 .. 3 write those of q = h
  if (h < 4) {
    for (int row = rg; row < nrows / AX_TOK; row += 4) {
      bf16_t* dst = vo + ((size_t)((f0 + row) * NB + nb) * 3 + 2) * 256 + (h * 16 + frow) * 4;
      dst[2] = 0;
      dst[3] = 0;
    }
  }
}

}  // namespace

extern "C" int64_t vx_audio_xattn_packed_bytes(int c, int frames) {
  // Kq: frames x (c / 32) x 3 KiB, VO: frames x (c / 16) x 1.5 KiB  -> 2 x 96 c bytes per frame
  return (int64_t)frames * c * 96;
}

extern "C" int vx_audio_xattn_supported(int c, int heads, int n_ctx, int rows_per_frame) {
  return heads == AX_HEADS && n_ctx == AX_TOK && c > 0 && (c % 320) == 0 && (c / heads) <= 256 && rows_per_frame > 0 &&
                 (rows_per_frame % 16) == 0
             ? 1
             : 0;
}

extern "C" int vx_audio_xattn_pack(const void* kv, int ldkv, const void* wq, const float* bq, const void* wo, int c, int heads,
                                   int n_ctx, int frames, void* kq, float* kq_colsum, float* kq_bias, void* vo, void* stream_) {
  VX_REQUIRE(kv != nullptr && wq != nullptr && wo != nullptr && kq != nullptr && kq_colsum != nullptr && kq_bias != nullptr &&
                 vo != nullptr, "vx_audio_xattn_pack: null pointer");
  VX_REQUIRE(vx_audio_xattn_supported(c, heads, n_ctx, 16), "vx_audio_xattn_pack: c=%d heads=%d n_ctx=%d (8 heads, 5 tokens, c %% 320 == 0)",
             c, heads, n_ctx);
  VX_REQUIRE(frames > 0 && ldkv >= 2 * c, "vx_audio_xattn_pack: frames=%d ldkv=%d", frames, ldkv);
  hipStream_t stream = (hipStream_t)stream_;
  const int d = c / heads;
  const float scale = 1.4426950408889634f / sqrtf((float)d);
  const unsigned fz = (unsigned)((frames + AXP_FR - 1) / AXP_FR);
  const size_t lds = (size_t)AXP_ROWS * d * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)axattn_pack_kq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 256 * 4);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)axattn_pack_vo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 256 * 4);
    if (e != hipSuccess) {
      vx_set_error("vx_audio_xattn_pack: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(axattn_pack_kq_kernel, dim3(AX_HEADS, c / 64, fz), dim3(256), lds, stream, (const bf16_t*)kv, ldkv,
                     (const bf16_t*)wq, c, d, scale, frames, (bf16_t*)kq);
  int rc = vx_check_launch("vx_audio_xattn_pack(kq)");
  if (rc) return rc;
  hipLaunchKernelGGL(axattn_pack_sums_kernel, dim3(frames, AX_SLOTS), dim3(256), 0, stream, (const bf16_t*)kv, ldkv, bq, c, d,
                     scale, (bf16_t*)kq, kq_colsum, kq_bias);
  rc = vx_check_launch("vx_audio_xattn_pack(sums)");
  if (rc) return rc;
  hipLaunchKernelGGL(axattn_pack_vo_kernel, dim3(AX_HEADS, c / 64, fz), dim3(256), lds, stream, (const bf16_t*)kv, ldkv,
                     (const bf16_t*)wo, c, d, frames, (bf16_t*)vo);
  return vx_check_launch("vx_audio_xattn_pack(vo)");
}

extern "C" int vx_audio_xattn(const vx_axattn_params* pp, void* stream_) {
  const vx_axattn_params& p = *pp;
  VX_REQUIRE(p.x != nullptr && p.out != nullptr && p.ln_stats != nullptr && p.kq != nullptr && p.kq_colsum != nullptr &&
                 p.kq_bias != nullptr && p.vo != nullptr && p.bias_o != nullptr, "vx_audio_xattn: null pointer");
  VX_REQUIRE(vx_audio_xattn_supported(p.c, AX_HEADS, AX_TOK, p.rows_per_frame), "vx_audio_xattn: c=%d rows_per_frame=%d", p.c,
             p.rows_per_frame);
  VX_REQUIRE(p.rows > 0 && (p.rows % p.rows_per_frame) == 0, "vx_audio_xattn: rows=%d is not whole frames of %d rows", p.rows,
             p.rows_per_frame);
  VX_REQUIRE((p.ldx % 8) == 0 && (p.ldo % 4) == 0 && p.ldx >= p.c && p.ldo >= p.c, "vx_audio_xattn: row strides");
  VX_REQUIRE(p.ln_stats_parts == 0 || p.ln_stats_parts == 2, "vx_audio_xattn: ln_stats_parts");
  VX_REQUIRE(p.row_stats_parts == 0 || p.row_stats_parts == 2, "vx_audio_xattn: row_stats_parts");
  hipStream_t stream = (hipStream_t)stream_;
  // 32 rows per wave where a FRAME alone gives every CU a wave (the 64x64 level and above: >= 4096 rows per frame); below
  // that the four waves of a workgroup share 16 rows (axattn_split_kernel).  The forms add the same products in different
  // orders, so the choice is a function of the per-frame shape only, never of how many frames share the launch: a CFG half, a
  // window or a frame shard computed alone has the bits of the batched call.  VX_AX_FORM=0 / 1 / 2 forces 32-row waves /
  // 16-row waves / the split (A/B knob).
  static int form_env = -2;
  if (form_env == -2) {
    const char* e = getenv("VX_AX_FORM");
    form_env = e ? atoi(e) : -1;
  }
  int form = (p.rows_per_frame % 32) == 0 && p.rows_per_frame >= 4096 ? 0 : 2;
  if (form_env >= 0 && form_env <= 2) form = form_env;
  if (form == 0 && (p.rows_per_frame % 32) != 0) form = 1;
  if (form == 0) {
    g_vx_last_kernel = "axattn_kernel<2>";
    hipLaunchKernelGGL(axattn_kernel<2>, dim3((unsigned)((p.rows + 127) / 128)), dim3(256), 0, stream, p);
  } else if (form == 1) {
    g_vx_last_kernel = "axattn_kernel<1>";
    hipLaunchKernelGGL(axattn_kernel<1>, dim3((unsigned)((p.rows + 63) / 64)), dim3(256), 0, stream, p);
  } else {
    g_vx_last_kernel = "axattn_split_kernel";
    hipLaunchKernelGGL(axattn_split_kernel, dim3((unsigned)(p.rows / 16)), dim3(256), 0, stream, p);
  }
  return vx_check_launch("vx_audio_xattn");
}
