// Attention kernels for gfx950 (see include/vexpress_hip.h): flash-style spatial attention on MFMA, temporal
// (frame-axis) attention, and cross-attention against a short key list.
//
// MFMA mapping used by both MFMA kernels (v_mfma_f32_16x16x32_bf16, lane = (i = lane&15, g = lane>>4)):
//   S^T[key, query] = K[key, :] . Q[query, :]      A = K fragment  (row = key i,  8 d-values at 32kk+8g)
//                                                   B = Q fragment  (col = query i, same d-values)
//                                                   C: reg r holds key 16kt+4g+r for query i
//   => every lane owns ONE query column: the softmax statistics are per-lane scalars and the rescale of the
//      output accumulator needs no cross-lane traffic; the row max/sum over keys is an in-lane reduction plus two
//      xor-shuffles over g.
//   O^T[dcol, query] = V^T[dcol, key] . P^T[key, query]   A = V^T fragment (row = dcol i, keys {4g..4g+3, 16+4g..})
//                                                          B = P^T fragment = the C registers of two S^T tiles,
//                                                              converted to bf16 in place (contraction order over
//                                                              keys is arbitrary, so no lane exchange is needed)
// V arrives pre-transposed ([.., head, dim, key], written by the QKV GEMM epilogue) so both LDS tiles are filled with
// 16-B row copies and read conflict-free: K as [d/32 slabs][64 keys][64 B] with chunk ^= (-(key>>2))&3, V^T as
// [dim rows][128 B] with chunk ^= (row>>1)&7.
#include "vx_common.h"
#include <stdio.h>
#include "../../include/vexpress_hip.h"

#include <stdlib.h>

#include <type_traits>

namespace {

struct AttnParams {
  const bf16_t* q; int ldq;
  const bf16_t* k; int ldk;
  const bf16_t* vt; int vt_pitch;
  bf16_t* out; int ldo;
  int batch, heads, n_q, n_kv, d, q_per_kv;
  float c;   // scale * log2(e)
  const float* kmax;   // [kv batches * heads] max key norm of the head slice (bounded-softmax variant of attn2), or null
};

__device__ __forceinline__ int k_lds_off(int slab, int key, int grp) {
  return slab * 4096 + key * 64 + ((grp ^ ((0 - (key >> 2)) & 3)) << 4);
}
__device__ __forceinline__ int v_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int KK, int DT, int QT, bool PREFETCH, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_kernel(const AttnParams p) {
  constexpr int NV = (DT + 1) / 2;   // V^T chunks per thread per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;
  char* vs = smem + KK * 4096;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int kvb = b / p.q_per_kv;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
  const bf16_t* __restrict__ kbase = p.k + (size_t)kvb * p.n_kv * p.ldk + h * p.d;
  const bf16_t* __restrict__ vbase = p.vt + (size_t)(kvb * p.heads + h) * p.d * p.vt_pitch;

  // Q fragments (B operand), loaded once
  uint4 qf[QT][KK];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int qrow = q0 + 16 * qt + i;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      int dcol = 32 * kk + 8 * g;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow < p.n_q && dcol < p.d)
        v = *reinterpret_cast<const uint4*>(p.q + (size_t)(b * p.n_q + qrow) * p.ldq + h * p.d + dcol);
      qf[qt][kk] = v;
    }
  }

  f32x4_t o[DT][QT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  uint4 rk[KK], rv[NV];
  auto load_k = [&](int t, int s) -> uint4 {
    int idx = tid + 256 * s;
    int key = idx / (4 * KK), c = idx % (4 * KK);
    int kg = t * 64 + key;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kg < p.n_kv && c * 8 < p.d) v = *reinterpret_cast<const uint4*>(kbase + (size_t)kg * p.ldk + c * 8);
    return v;
  };
  auto store_k = [&](int s, const uint4& v) {
    int idx = tid + 256 * s;
    int key = idx / (4 * KK), c = idx % (4 * KK);
    *reinterpret_cast<uint4*>(ks + k_lds_off(c >> 2, key, c & 3)) = v;
  };
  auto load_v = [&](int t, int s) -> uint4 {
    int idx = tid + 256 * s;
    int row = idx >> 3, j = idx & 7;
    int kg = t * 64 + j * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < p.d && kg < p.n_kv) v = *reinterpret_cast<const uint4*>(vbase + (size_t)row * p.vt_pitch + kg);
    return v;
  };
  auto store_v = [&](int s, const uint4& v) {
    int idx = tid + 256 * s;
    int row = idx >> 3, j = idx & 7;
    if (row < DT * 16) *reinterpret_cast<uint4*>(vs + v_lds_off(row, j)) = v;
  };

  const int n_tiles = (p.n_kv + 63) >> 6;
  if constexpr (PREFETCH) {
#pragma unroll
    for (int s = 0; s < KK; ++s) rk[s] = load_k(0, s);
#pragma unroll
    for (int s = 0; s < NV; ++s) rv[s] = load_v(0, s);
#pragma unroll
    for (int s = 0; s < KK; ++s) store_k(s, rk[s]);
#pragma unroll
    for (int s = 0; s < NV; ++s) store_v(s, rv[s]);
    __syncthreads();
  }

  for (int t = 0; t < n_tiles; ++t) {
    if constexpr (PREFETCH) {
      if (t + 1 < n_tiles) {
#pragma unroll
        for (int s = 0; s < KK; ++s) rk[s] = load_k(t + 1, s);
#pragma unroll
        for (int s = 0; s < NV; ++s) rv[s] = load_v(t + 1, s);
      }
    } else {
      __syncthreads();
#pragma unroll 4
      for (int s = 0; s < KK; ++s) store_k(s, load_k(t, s));
#pragma unroll 4
      for (int s = 0; s < NV; ++s) store_v(s, load_v(t, s));
      __syncthreads();
    }

    // ---- S^T = K Q^T
    f32x4_t s_[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s_[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        uint4 kf = *reinterpret_cast<const uint4*>(ks + k_lds_off(kk, 16 * kt + i, g));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s_[kt][qt] = mfma16(kf, qf[qt][kk], s_[kt][qt]);
      }
    }
    if ((t + 1) * 64 > p.n_kv) {   // ragged last tile: keys beyond n_kv never contribute
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (t * 64 + 16 * kt + 4 * g + r >= p.n_kv) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s_[kt][qt][r] = -INFINITY;
          }
        }
    }
    // ---- online softmax (per-lane query column)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = s_[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s_[kt][qt][r]);
      mx = wave_rows_max(mx);
      float mnew = fmaxf(m_run[qt], mx * p.c);
      float alpha = __builtin_amdgcn_exp2f(m_run[qt] - mnew);
      m_run[qt] = mnew;
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(s_[kt][qt][r], p.c, -mnew));
          rs += pv;
          s_[kt][qt][r] = pv;
        }
      l_run[qt] = l_run[qt] * alpha + rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        o[dt][qt][0] *= alpha; o[dt][qt][1] *= alpha; o[dt][qt][2] *= alpha; o[dt][qt][3] *= alpha;
      }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int ks_ = 0; ks_ < 2; ++ks_) {
      uint4 pb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4_t& a = s_[2 * ks_][qt];
        const f32x4_t& c2 = s_[2 * ks_ + 1][qt];
        pb[qt] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(c2[0], c2[1]),
                            pack_bf16x2(c2[2], c2[3]));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        int row = 16 * dt + i;
        int j1 = 4 * ks_ + (g >> 1);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vs + v_lds_off(row, j1) + (g & 1) * 8);
        const uint2 v2 = *reinterpret_cast<const uint2*>(vs + v_lds_off(row, j1 + 2) + (g & 1) * 8);
        uint4 vf = make_uint4(v1.x, v1.y, v2.x, v2.y);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = mfma16(vf, pb[qt], o[dt][qt]);
      }
    }
    if constexpr (PREFETCH) {
      __syncthreads();
      if (t + 1 < n_tiles) {
#pragma unroll
        for (int s = 0; s < KK; ++s) store_k(s, rk[s]);
#pragma unroll
        for (int s = 0; s < NV; ++s) store_v(s, rv[s]);
      }
      __syncthreads();
    }
  }

  // ---- normalise and store: lane holds 4 consecutive d-columns of one query
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    l = wave_xor_sum(l, 16);
    l = wave_xor_sum(l, 32);
    float inv = 1.0f / l;
    int qrow = q0 + 16 * qt + i;
    if (qrow >= p.n_q) continue;
    bf16_t* orow = p.out + (size_t)(b * p.n_q + qrow) * p.ldo + h * p.d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      int dcol = 16 * dt + 4 * g;
      if (dcol < p.d) {
        uint2 w;
        w.x = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
        w.y = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
        *reinterpret_cast<uint2*>(orow + dcol) = w;
      }
    }
  }
}

template <int KK, int DT, int QT, bool PREFETCH, int MINW = 2>
int launch_attn(const AttnParams& p, hipStream_t stream) {
  constexpr int smem = KK * 4096 + DT * 2048;
  auto kern = attn_kernel<KK, DT, QT, PREFETCH, MINW>;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vx_set_error("vx_attention: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid(ceil_div(p.n_q, 64 * QT), p.batch * p.heads);
  static char sym[64] = "";
  if (!sym[0]) snprintf(sym, sizeof(sym), "attn_kernel<%d, %d, %d, %s, %d>", KK, DT, QT, PREFETCH ? "true" : "false", MINW);
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
  return vx_check_launch("vx_attention");
}

// -------------------------------------------------------------------------------------------------------------
// attn2_kernel: the same MFMA mapping, re-scheduled around the real bound of the small-head-dim levels.  At d = 40
// (64x64 level: 4096 x 4096 scores per head) a 64-key tile costs a wave 28 MFMAs (448 cycles) but ~185 VALU
// instructions for the online softmax (32 of them v_exp_f32): the kernel is VALU-bound, and the old body ran the two
// streams back to back.  Changes:
//   * the per-tile VALU overhead that is not softmax is gone: key masking only in a peeled last tile, staging
//     addresses and validity precomputed (unconditional loads: predicated ones made hipcc serialise them);
//   * row sums come out of the PV MFMA: row `d` of the V^T tile is all ones (head dims that are not multiples of 16
//     have spare rows), so O^T[d][q] = sum_k P[q][k] of exactly the bf16 P values that multiply V; no VALU adds;
//   * the accumulator rescale (and its multiplies) is skipped, wave-uniformly, whenever no running max of the wave
//     grew in this tile (alpha == 1 for every lane): after the first few tiles that is the common case;
//   * row max through v_max3_f32 (fmaxf nests fold to it).
// LDS: single K / V^T buffers; the next tile waits in registers while this one is multiplied (T14 split).
//
// BOUND (p.kmax != null): softmax is invariant to the per-query shift, so the shift need not be the running row max -
// any m_i >= max_j s_ij keeps every exponent <= 0.  Cauchy-Schwarz gives one for free: s_ij <= |q_i| max_j |k_j|.
// With m_i = c |q_i| Kmax fixed for the whole key loop (Kmax per (batch, head) from vx_key_norm_max, |q_i| from the Q
// fragments) the per-tile row max (16 v_max3/v_max + 2 cross-row exchanges per query tile, all on the dependency chain
// in front of the exponentials), the running-max update, alpha and the accumulator rescale disappear: a tile is
// 32 fma + 32 exp + 16 cvt.  The price is range: a query whose true max lies more than ~100 (log2 units) under its
// bound has all its probabilities flushed towards zero.  Detected at the end (row sum < 2^-100), block-uniformly, and
// the block then recomputes with the exact online softmax - never a wrong result, at worst the old speed plus one
// wasted pass (outlier-norm keys in trained weights can trigger that; random-init weights never do).
template <int KK, int DT, int QT, bool ONES, bool BOUND>
__global__ __launch_bounds__(256, 2) void attn2_kernel(const AttnParams p) {
  constexpr int NV = (DT + 1) / 2;   // V^T chunks per thread per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;
  char* vs = smem + KK * 4096;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // 1-D grid, XCD-aware: blocks are dealt to the 8 XCDs round-robin, so logical ids that are adjacent (all query
  // blocks of one (batch, head), then the next head) are mapped onto ONE XCD and its L2 serves the K / V^T re-reads.
  const int nqb = (p.n_q + 64 * QT - 1) / (64 * QT);
  int lid;
  {
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int bh = lid / nqb, qb = lid - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int kvb = b / p.q_per_kv;
  const int q0 = qb * (64 * QT) + wave * (16 * QT);
  const bf16_t* __restrict__ kbase = p.k + (size_t)kvb * p.n_kv * p.ldk + h * p.d;
  const bf16_t* __restrict__ vbase = p.vt + (size_t)(kvb * p.heads + h) * p.d * p.vt_pitch;

  uint4 qf[QT][KK];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int qrow = q0 + 16 * qt + i;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      int dcol = 32 * kk + 8 * g;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow < p.n_q && dcol < p.d)
        v = *reinterpret_cast<const uint4*>(p.q + (size_t)(b * p.n_q + qrow) * p.ldq + h * p.d + dcol);
      qf[qt][kk] = v;
    }
  }
  f32x4_t o[DT][QT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  float mfix[QT];   // BOUND: c |q_i| Kmax, identical in the 4 lanes of a query column
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mfix[qt] = 0.f;
    if (BOUND) {
      float ss = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        float f[8];
        unpack_bf16x8(qf[qt][kk], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
      }
      ss = wave_xor_sum(ss, 16);
      ss = wave_xor_sum(ss, 32);
      mfix[qt] = p.c * sqrtf(ss) * p.kmax[kvb * p.heads + h] - VX_P_HEADROOM;
    }
  }

  // ---- staging coordinates (loop-invariant): thread -> (key, 16-B channel chunk) of the K tile and (dim row, 8-key
  // chunk) of the V^T tile.  Chunks beyond the head dim are zero-filled ONCE (the single LDS buffers are never
  // overwritten there) and row d of V^T is set to ones once.  In the loop every thread loads unconditionally (threads
  // of a padding chunk re-read chunk 0 / row 0 and just do not store): predicated loads would make hipcc wait for the
  // previous tile's loads before issuing the next ones.
  uint32_t kptr[KK];   // element offsets from kbase / vbase (the K / V^T of one (batch, head) span < 4 Gi elements)
  char* klds[KK];
  bool kval[KK];
#pragma unroll
  for (int s = 0; s < KK; ++s) {
    const int idx = tid + 256 * s;
    const int key = idx / (4 * KK), c = idx % (4 * KK);
    kval[s] = c * 8 < p.d;
    kptr[s] = (uint32_t)key * (uint32_t)p.ldk + (kval[s] ? c * 8 : 0);
    klds[s] = ks + k_lds_off(c >> 2, key, c & 3);
    if (!kval[s]) *reinterpret_cast<uint4*>(klds[s]) = make_uint4(0, 0, 0, 0);
  }
  uint32_t vptr[NV];
  char* vlds[NV];
  bool vval[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int idx = tid + 256 * s;
    const int row = idx >> 3, j = idx & 7;
    vval[s] = row < p.d;
    vptr[s] = (uint32_t)(vval[s] ? row : 0) * (uint32_t)p.vt_pitch + j * 8;
    vlds[s] = vs + v_lds_off(row, j);
    if (!vval[s] && row < DT * 16)
      *reinterpret_cast<uint4*>(vlds[s]) = (ONES && row == p.d)
                                               ? make_uint4(VX_E16_ONE2, VX_E16_ONE2, VX_E16_ONE2, VX_E16_ONE2)
                                               : make_uint4(0, 0, 0, 0);
  }
  const uint32_t kstep = 64u * (uint32_t)p.ldk;   // elements per key tile
  const int n_tiles = (p.n_kv + 63) >> 6;
  const bool ragged = (p.n_kv & 63) != 0;    // the last tile has keys beyond n_kv

  uint4 rk[KK], rv[NV];
  // tile `t` -> registers.  `edge`: the tile reaches beyond n_kv; those keys re-read the last valid key / the first
  // key chunk instead (finite values: their scores are masked to -inf, so P = 0 exactly and 0 * V stays 0).
  // (value-returning helpers + plain unrolled loops at the call sites: arrays written through a by-reference lambda
  // ended up in scratch memory)
  auto load_k1 = [&](int t, int s, bool edge) -> uint4 {
    uint32_t off = kptr[s] + (uint32_t)t * kstep;
    if (edge) {
      const int key = (tid + 256 * s) / (4 * KK);
      const int over = t * 64 + key - (p.n_kv - 1);
      if (over > 0) off -= (uint32_t)over * (uint32_t)p.ldk;
    }
    return *reinterpret_cast<const uint4*>(kbase + off);
  };
  auto load_v1 = [&](int t, int s, bool edge) -> uint4 {
    uint32_t off = vptr[s] + (uint32_t)t * 64u;
    if (edge) {
      const int j = (tid + 256 * s) & 7;
      if (t * 64 + j * 8 >= p.n_kv) off = vptr[s] - j * 8;   // chunk 0 of the row
    }
    return *reinterpret_cast<const uint4*>(vbase + off);
  };
#define VX_ATTN2_LOAD(t_, edge_)                                      \
  do {                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < KK; ++s_) rk[s_] = load_k1((t_), s_, (edge_)); \
    _Pragma("unroll") for (int s_ = 0; s_ < NV; ++s_) rv[s_] = load_v1((t_), s_, (edge_)); \
  } while (0)
#define VX_ATTN2_STORE()                                              \
  do {                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < KK; ++s_)                 \
      if (kval[s_]) *reinterpret_cast<uint4*>(klds[s_]) = rk[s_];     \
    _Pragma("unroll") for (int s_ = 0; s_ < NV; ++s_)                 \
      if (vval[s_]) *reinterpret_cast<uint4*>(vlds[s_]) = rv[s_];     \
  } while (0)
  auto qk = [&](f32x4_t (&s_)[4][QT]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s_[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        uint4 kf = *reinterpret_cast<const uint4*>(ks + k_lds_off(kk, 16 * kt + i, g));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s_[kt][qt] = mfma16(kf, qf[qt][kk], s_[kt][qt]);
      }
    }
  };

  // one key tile.  MASK: tile t reaches beyond n_kv.  FIXED: the bounded-softmax body (see BOUND above).
  auto step = [&](auto mask_c, auto fixed_c, const int t) {
    constexpr bool MASK = decltype(mask_c)::value;
    constexpr bool FIXED = decltype(fixed_c)::value;
    // registers <- K(t+1), V(t+1)   (written to LDS after this tile's MFMAs: T14 issue-early / write-late)
    if (t + 1 < n_tiles) {
      const bool edge = ragged && t + 1 == n_tiles - 1;
      VX_ATTN2_LOAD(t + 1, edge);
    }
    f32x4_t cur[4][QT];
    qk(cur);
    if (MASK) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (t * 64 + 16 * kt + 4 * g + r >= p.n_kv) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) cur[kt][qt][r] = -INFINITY;
          }
        }
    }
    if constexpr (FIXED) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(fmaf(cur[kt][qt][r], p.c, -mfix[qt]));
            if (!ONES) rs += pv;
            cur[kt][qt][r] = pv;
          }
        if (!ONES) l_run[qt] += rs;
      }
    } else {
    float alpha[QT];
    bool grew = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = fmaxf(fmaxf(cur[0][qt][0], cur[0][qt][1]), cur[0][qt][2]);
      mx = fmaxf(fmaxf(mx, cur[0][qt][3]), cur[1][qt][0]);
      mx = fmaxf(fmaxf(mx, cur[1][qt][1]), cur[1][qt][2]);
      mx = fmaxf(fmaxf(mx, cur[1][qt][3]), cur[2][qt][0]);
      mx = fmaxf(fmaxf(mx, cur[2][qt][1]), cur[2][qt][2]);
      mx = fmaxf(fmaxf(mx, cur[2][qt][3]), cur[3][qt][0]);
      mx = fmaxf(fmaxf(mx, cur[3][qt][1]), cur[3][qt][2]);
      mx = fmaxf(mx, cur[3][qt][3]);
      mx = wave_rows_max(mx);
      const float mnew = fmaxf(m_run[qt], mx * p.c);
      grew |= mnew > m_run[qt];
      alpha[qt] = __builtin_amdgcn_exp2f(m_run[qt] - mnew);
      m_run[qt] = mnew;
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(cur[kt][qt][r], p.c, -mnew));
          if (!ONES) rs += pv;
          cur[kt][qt][r] = pv;
        }
      if (!ONES) l_run[qt] = l_run[qt] * alpha[qt] + rs;   // (partial over g; reduced at the end)
    }
    if (__any(grew)) {   // wave-uniform: some running max moved -> rescale the accumulators (and the sums in them)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          o[dt][qt][0] *= alpha[qt]; o[dt][qt][1] *= alpha[qt]; o[dt][qt][2] *= alpha[qt]; o[dt][qt][3] *= alpha[qt];
        }
    }
    }   // !FIXED
    // ---- O^T += V^T P^T
#pragma unroll
    for (int ks_ = 0; ks_ < 2; ++ks_) {
      uint4 pb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4_t& a = cur[2 * ks_][qt];
        const f32x4_t& c2 = cur[2 * ks_ + 1][qt];
        pb[qt] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(c2[0], c2[1]),
                            pack_bf16x2(c2[2], c2[3]));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        int row = 16 * dt + i;
        int j1 = 4 * ks_ + (g >> 1);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vs + v_lds_off(row, j1) + (g & 1) * 8);
        const uint2 v2 = *reinterpret_cast<const uint2*>(vs + v_lds_off(row, j1 + 2) + (g & 1) * 8);
        uint4 vf = make_uint4(v1.x, v1.y, v2.x, v2.y);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = mfma16(vf, pb[qt], o[dt][qt]);
      }
    }
    // LDS <- K(t+1), V(t+1)
    __syncthreads();
    if (t + 1 < n_tiles) VX_ATTN2_STORE();
    __syncthreads();
  };

  // the whole key loop (prologue: K(0), V(0) -> LDS; every step ends with a block barrier, so LDS is free on entry)
  auto run = [&](auto fixed_c) {
    VX_ATTN2_LOAD(0, n_tiles == 1 && ragged);
    VX_ATTN2_STORE();
    __syncthreads();
    const int n_plain = ragged ? n_tiles - 1 : n_tiles;   // tiles that need no key masking
    for (int t = 0; t < n_plain; ++t) step(std::false_type{}, fixed_c, t);
    if (ragged) step(std::true_type{}, fixed_c, n_tiles - 1);
  };
  // softmax denominator of query tile qt (every lane of the query's column gets it)
  auto row_sum = [&](int qt) -> float {
    float l;
    if (ONES) {
      // O^T[d][query i] lives in fragment d / 16, lane group (d % 16) / 4, register d % 4
      const int dd = p.d;
      float mine = 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * dt + r == dd - 4 * ((dd & 15) >> 2)) mine = o[dt][qt][r];
      l = __shfl(mine, i + 16 * ((dd & 15) >> 2), 64);
    } else {
      l = l_run[qt];
      l = wave_xor_sum(l, 16);
      l = wave_xor_sum(l, 32);
    }
    return l;
  };

  if constexpr (BOUND) {
    run(std::true_type{});
    bool bad = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float l = row_sum(qt);
      bad |= (q0 + 16 * qt + i < p.n_q) && !(l >= VX_P_MIN_ROWSUM(p.n_kv));   // (bf16: 2^-100); also catches NaN
    }
    if (__syncthreads_or(bad)) {   // block-uniform (the K / V^T staging is shared by the 4 waves): exact recompute
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      run(std::false_type{});
    }
  } else {
    run(std::false_type{});
  }

  // ---- normalise and store: lane holds 4 consecutive d-columns of one query
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = row_sum(qt);
    float inv = 1.0f / l;
    int qrow = q0 + 16 * qt + i;
    if (qrow >= p.n_q) continue;
    bf16_t* orow = p.out + (size_t)(b * p.n_q + qrow) * p.ldo + h * p.d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      int dcol = 16 * dt + 4 * g;
      if (dcol < p.d) {
        uint2 w;
        w.x = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
        w.y = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
        *reinterpret_cast<uint2*>(orow + dcol) = w;
      }
    }
  }
}

#undef VX_ATTN2_LOAD
#undef VX_ATTN2_STORE

template <int KK, int DT, int QT, bool ONES, bool BOUND>
int launch_attn2(const AttnParams& p, hipStream_t stream) {
  constexpr int smem = KK * 4096 + DT * 2048;
  auto kern = attn2_kernel<KK, DT, QT, ONES, BOUND>;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vx_set_error("vx_attention: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)((long)ceil_div(p.n_q, 64 * QT) * p.batch * p.heads));
  static char sym[64] = "";
  if (!sym[0])
    snprintf(sym, sizeof(sym), "attn2_kernel<%d, %d, %d, %s, %s>", KK, DT, QT, ONES ? "true" : "false", BOUND ? "true" : "false");
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
  return vx_check_launch("vx_attention");
}

// -------------------------------------------------------------------------------------------- temporal attention
struct TemporalParams {
  const bf16_t* qkv; int ldqkv;
  bf16_t* out; int ldo;
  int b, f, hw, heads, d;
  float c;
};

// WPB waves per block, consecutive heads of one pixel: with WPB = heads = 8 a block reads whole 128-byte lines of the
// token rows (a head's q / k / v slice is d * 2 = 80 bytes at d = 40: with 4 heads per block the line in the middle of a
// pixel's 8 heads was fetched by two blocks on two XCDs, i.e. two L2s - 1.6x the traffic of an HBM-bound kernel)
template <int KK, int DT, int FT, int WPB>
__global__ __launch_bounds__(64 * WPB) void temporal_attn_kernel(const TemporalParams p) {
  constexpr int VP = 32 * KK + 8;   // V tile pitch in elements (pad breaks the power-of-two stride)
  __shared__ __attribute__((aligned(16))) bf16_t vsm[WPB][16 * FT][VP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long total = (long)p.b * p.hw * p.heads;
  long wg = (long)blockIdx.x * WPB + wave;
  const bool live = wg < total;
  if (!live) wg = total - 1;
  const int h = (int)(wg % p.heads);
  const int pix = (int)((wg / p.heads) % p.hw);
  const int bb = (int)(wg / ((long)p.heads * p.hw));
  const int C = p.heads * p.d;
  auto row_ptr = [&](int fr) { return p.qkv + ((size_t)(bb * p.f + fr) * p.hw + pix) * p.ldqkv + h * p.d; };

  uint4 qf[FT][KK], kf[FT][KK];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    int fr = 16 * ft + i;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      int dcol = 32 * kk + 8 * g;
      uint4 qv = make_uint4(0, 0, 0, 0), kv = qv;
      if (fr < p.f && dcol < p.d) {
        const bf16_t* rp = row_ptr(fr);
        qv = *reinterpret_cast<const uint4*>(rp + dcol);
        kv = *reinterpret_cast<const uint4*>(rp + C + dcol);
      }
      qf[ft][kk] = qv;
      kf[ft][kk] = kv;
    }
  }
  // stage V rows (natural [frame][d]) into this wave's LDS tile; rows >= f and columns >= d are zero
#pragma unroll
  for (int s = 0; s < FT * KK; ++s) {
    int idx = lane + 64 * s;
    int fr = idx / (4 * KK), c = idx % (4 * KK);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (fr < p.f && c * 8 < p.d) v = *reinterpret_cast<const uint4*>(row_ptr(fr) + 2 * C + c * 8);
    *reinterpret_cast<uint4*>(&vsm[wave][fr][c * 8]) = v;
  }
  __syncthreads();

  f32x4_t s_[FT][FT];
#pragma unroll
  for (int kt = 0; kt < FT; ++kt)
#pragma unroll
    for (int qt = 0; qt < FT; ++qt) {
      f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) a = mfma16(kf[kt][kk], qf[qt][kk], a);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * g + r >= p.f) a[r] = -INFINITY;
      s_[kt][qt] = a;
    }
  float inv_l[FT];
#pragma unroll
  for (int qt = 0; qt < FT; ++qt) {
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < FT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s_[kt][qt][r]);
    mx = wave_xor_max(mx, 16);
    mx = wave_xor_max(mx, 32);
    float ms = mx * p.c, rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < FT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __builtin_amdgcn_exp2f(fmaf(s_[kt][qt][r], p.c, -ms));
        rs += pv;
        s_[kt][qt][r] = pv;
      }
    rs = wave_xor_sum(rs, 16);
    rs = wave_xor_sum(rs, 32);
    inv_l[qt] = 1.0f / rs;
  }
  uint4 pb[FT];
#pragma unroll
  for (int qt = 0; qt < FT; ++qt) {
    const f32x4_t& a = s_[0][qt];
    uint32_t hi0 = 0, hi1 = 0;
    if (FT == 2) {
      const f32x4_t& c2 = s_[FT - 1][qt];
      hi0 = pack_bf16x2(c2[0], c2[1]);
      hi1 = pack_bf16x2(c2[2], c2[3]);
    }
    pb[qt] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), hi0, hi1);
  }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    int dcol = 16 * dt + i;
    uint32_t w[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
      if (half < FT && dcol < 32 * KK) {
        int fr0 = 16 * half + 4 * g;
        e0 = vsm[wave][fr0 + 0][dcol]; e1 = vsm[wave][fr0 + 1][dcol];
        e2 = vsm[wave][fr0 + 2][dcol]; e3 = vsm[wave][fr0 + 3][dcol];
      }
      w[2 * half + 0] = (uint32_t)e0 | ((uint32_t)e1 << 16);
      w[2 * half + 1] = (uint32_t)e2 | ((uint32_t)e3 << 16);
    }
    uint4 vf = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
    for (int qt = 0; qt < FT; ++qt) {
      f32x4_t acc = mfma16(vf, pb[qt], f32x4_t{0.f, 0.f, 0.f, 0.f});
      int fr = 16 * qt + i;
      int dc = 16 * dt + 4 * g;
      if (live && fr < p.f && dc < p.d) {
        uint2 st;
        st.x = pack_bf16x2(acc[0] * inv_l[qt], acc[1] * inv_l[qt]);
        st.y = pack_bf16x2(acc[2] * inv_l[qt], acc[3] * inv_l[qt]);
        *reinterpret_cast<uint2*>(p.out + ((size_t)(bb * p.f + fr) * p.hw + pix) * p.ldo + h * p.d + dc) = st;
      }
    }
  }
}

template <int KK, int DT>
int launch_temporal(const TemporalParams& p, hipStream_t stream) {
  long waves = (long)p.b * p.hw * p.heads;
  static int wpb_env = -1;   // VX_TEMPORAL_WPB = 4 | 8 (A/B knob; default: 8 when the head count allows)
  if (wpb_env < 0) {
    const char* e = getenv("VX_TEMPORAL_WPB");
    wpb_env = e ? atoi(e) : 0;
  }
  const bool wide = wpb_env ? wpb_env == 8 : (p.heads % 8) == 0;
  static char sym[4][56];
  auto name = [&](int slot, int ft, int wpb) {
    if (!sym[slot][0]) snprintf(sym[slot], sizeof(sym[slot]), "temporal_attn_kernel<%d, %d, %d, %d>", KK, DT, ft, wpb);
    g_vx_last_kernel = sym[slot];
  };
  if (wide && (p.heads % 8) == 0 && KK * (p.f <= 16 ? 1 : 2) <= 6) {   // LDS: 8 x 16 FT x (32 KK + 8) x 2 B <= 64 KiB
    dim3 grid((unsigned)((waves + 7) / 8));
    name(p.f <= 16 ? 0 : 1, p.f <= 16 ? 1 : 2, 8);
    if (p.f <= 16) hipLaunchKernelGGL((temporal_attn_kernel<KK, DT, 1, 8>), grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((temporal_attn_kernel<KK, DT, 2, 8>), grid, dim3(512), 0, stream, p);
    return vx_check_launch("vx_temporal_attention");
  }
  dim3 grid((unsigned)((waves + 3) / 4));
  name(p.f <= 16 ? 2 : 3, p.f <= 16 ? 1 : 2, 4);
  if (p.f <= 16) hipLaunchKernelGGL((temporal_attn_kernel<KK, DT, 1, 4>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((temporal_attn_kernel<KK, DT, 2, 4>), grid, dim3(256), 0, stream, p);
  return vx_check_launch("vx_temporal_attention");
}

// -------------------------------------------------------------------------------------- short key-list attention
constexpr int SKV_MAX = 16;

__global__ __launch_bounds__(256) void small_kv_attn_kernel(const bf16_t* __restrict__ q, int ldq,
                                                            const bf16_t* __restrict__ kv, int ldkv, int v_off,
                                                            bf16_t* __restrict__ out, int ldo, int n_q, int n_kv,
                                                            int heads, int d, float c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* kvs = reinterpret_cast<bf16_t*>(smem);   // [n_kv][2][C]
  const int C = heads * d;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int cch = C >> 3;
  for (int idx = tid; idx < n_kv * 2 * cch; idx += 256) {
    int j = idx / (2 * cch), rem = idx % (2 * cch);
    int which = rem / cch, ch = (rem % cch) * 8;
    const bf16_t* src = kv + ((size_t)b * n_kv + j) * ldkv + (which ? v_off : 0) + ch;
    *reinterpret_cast<uint4*>(kvs + ((size_t)(j * 2 + which)) * C + ch) = *reinterpret_cast<const uint4*>(src);
  }
  __syncthreads();
  const long item = (long)blockIdx.x * 256 + tid;   // (token, head), head fastest
  if (item >= (long)n_q * heads) return;
  const int h = (int)(item % heads);
  const int tok = (int)(item / heads);
  const bf16_t* qp = q + ((size_t)b * n_q + tok) * ldq + h * d;
  float sc[SKV_MAX];
#pragma unroll
  for (int j = 0; j < SKV_MAX; ++j) sc[j] = 0.f;
  for (int ch = 0; ch < d; ch += 8) {
    float qv[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(qp + ch), qv);
#pragma unroll
    for (int j = 0; j < SKV_MAX; ++j) {
      if (j < n_kv) {
        float kvv[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(kvs + (size_t)(j * 2) * C + h * d + ch), kvv);
#pragma unroll
        for (int e = 0; e < 8; ++e) sc[j] = fmaf(qv[e], kvv[e], sc[j]);
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SKV_MAX; ++j)
    if (j < n_kv) mx = fmaxf(mx, sc[j]);
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < SKV_MAX; ++j) {
    float pv = (j < n_kv) ? __builtin_amdgcn_exp2f((sc[j] - mx) * c) : 0.f;
    sc[j] = pv;
    den += pv;
  }
  const float inv = 1.0f / den;
  bf16_t* op = out + ((size_t)b * n_q + tok) * ldo + h * d;
  for (int ch = 0; ch < d; ch += 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SKV_MAX; ++j) {
      if (j < n_kv) {
        float vv[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(kvs + (size_t)(j * 2 + 1) * C + h * d + ch), vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(sc[j], vv[e], acc[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    *reinterpret_cast<uint4*>(op + ch) = pack_bf16x8(acc);
  }
}

// max over the keys of one (kv batch, head) of the Euclidean norm of the key's head slice (fp32): the Kmax of the
// bounded-softmax attention.  One block per (kv batch, head); a thread walks keys tid, tid + 256, ... with 16-B loads.
__global__ __launch_bounds__(256) void key_norm_max_kernel(const bf16_t* __restrict__ k, int ldk, int heads, int n_kv,
                                                           int d, float* __restrict__ out) {
  __shared__ float red[256];
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const bf16_t* base = k + (size_t)b * n_kv * ldk + h * d;
  const int chunks = d >> 3;
  float best = 0.f;
  for (int key = threadIdx.x; key < n_kv; key += 256) {
    const bf16_t* row = base + (size_t)key * ldk;
    float ss = 0.f;
    for (int c = 0; c < chunks; ++c) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(row + 8 * c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
    }
    best = fmaxf(best, ss);
  }
  red[threadIdx.x] = best;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[bh] = sqrtf(red[0]);
}

// The same table for EIGHT heads from whole rows (round 6).  The kernel above gives a thread one key's head slice: a wave's
// load touches 64 different rows, and the 32 KB L1 must hold every line until the slice's other chunks are read - at 4 waves
// per CU that works (2 TB/s), with more waves or more keys per thread in flight it thrashes (1.3 TB/s: profiles/r06w_*,
// r06x_*).  Here a wave reads 8 keys x 8 heads: lane = (key l >> 3, head l & 7), the five 16-byte chunks of its slice back to
// back, so a wave consumes eight whole rows (5 KB) at a time; a block = 256 keys of one kv batch, all heads; the block maxima
// meet in the table through an integer atomic maximum (the table is zeroed first; squared norms are non-negative floats, whose
// bit patterns order like the values, and a maximum does not depend on the order: same bits as the kernel above).
constexpr int KNM8_KEYS = 256;
__global__ __launch_bounds__(256) void key_norm_max8_kernel(const bf16_t* __restrict__ k, int ldk, int n_kv, int d,
                                                            float* __restrict__ out) {
  __shared__ float red[4][8];
  const int b = blockIdx.y, key0 = blockIdx.x * KNM8_KEYS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane & 7;
  const bf16_t* base = k + (size_t)b * n_kv * ldk + h * d;
  const int chunks = d >> 3;
  float best = 0.f;
  for (int kk = wave * 8 + (lane >> 3); kk < KNM8_KEYS; kk += 32) {
    const int key = min(key0 + kk, n_kv - 1);                     // (clamped: a repeated key does not change the maximum)
    const bf16_t* row = base + (size_t)key * ldk;
    float ss = 0.f;
    for (int c = 0; c < chunks; ++c) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(row + 8 * c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
    }
    best = fmaxf(best, ss);
  }
  best = fmaxf(best, __shfl_xor(best, 8, 64));
  best = fmaxf(best, __shfl_xor(best, 16, 64));
  best = fmaxf(best, __shfl_xor(best, 32, 64));
  if (lane < 8) red[wave][lane] = best;
  __syncthreads();
  if (threadIdx.x < 8) {
    const float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
    atomicMax(reinterpret_cast<unsigned int*>(out) + b * 8 + threadIdx.x, __float_as_uint(sqrtf(m)));
  }
}

int attention_impl(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out, int ldo,
                   int batch, int heads, int n_q, int n_kv, int head_dim, int q_per_kv, float scale,
                   const float* key_norm_max, hipStream_t stream);

}  // namespace

// vx_attn3.hip: the DMA-ring kernel for d = 40 with a key-norm table
int vx_attn3_variant();
int vx_attn3_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out, int ldo,
                    int batch, int heads, int n_q, int n_kv, int q_per_kv, float c, const float* kmax,
                    hipStream_t stream);

extern "C" int vx_key_norm_max(const void* k, int ldk, int kv_batches, int heads, int n_kv, int head_dim, float* out,
                               void* stream) {
  VX_REQUIRE(k && out, "vx_key_norm_max: null pointer");
  VX_REQUIRE(kv_batches > 0 && heads > 0 && n_kv > 0 && head_dim > 0 && (head_dim % 8) == 0 && (ldk % 8) == 0,
             "vx_key_norm_max: bad sizes (head_dim=%d ldk=%d)", head_dim, ldk);
  static int v8 = -1;
  if (v8 < 0) {
    const char* e = getenv("VX_KNM8");            // A/B knob: 0 = always the one-block-per-(batch, head) kernel
    v8 = !(e && atoi(e) == 0);
  }
  if (v8 && heads == 8 && n_kv >= 2 * KNM8_KEYS && kv_batches <= 65535) {
    // (short key sets - the 8x8 level, the audio tokens - keep the plain kernel: one block each is all they need)
    if (hipMemsetAsync(out, 0, (size_t)kv_batches * heads * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      vx_set_error("vx_key_norm_max: hipMemsetAsync failed");
      return VX_ERR_HIP;
    }
    hipLaunchKernelGGL(key_norm_max8_kernel, dim3((n_kv + KNM8_KEYS - 1) / KNM8_KEYS, kv_batches), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)k, ldk, n_kv, head_dim, out);
    return vx_check_launch("vx_key_norm_max");
  }
  hipLaunchKernelGGL(key_norm_max_kernel, dim3(kv_batches * heads), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)k, ldk, heads, n_kv, head_dim, out);
  return vx_check_launch("vx_key_norm_max");
}

extern "C" int vx_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out,
                            int ldo, int batch, int heads, int n_q, int n_kv, int head_dim, int q_per_kv,
                            float scale, void* stream) {
  return attention_impl(q, ldq, k, ldk, vt, vt_pitch, out, ldo, batch, heads, n_q, n_kv, head_dim, q_per_kv, scale,
                        nullptr, (hipStream_t)stream);
}

extern "C" int vx_attention_bounded(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch,
                                    void* out, int ldo, int batch, int heads, int n_q, int n_kv, int head_dim,
                                    int q_per_kv, float scale, const float* key_norm_max, void* stream) {
  VX_REQUIRE(key_norm_max != nullptr, "vx_attention_bounded: null key_norm_max");
  return attention_impl(q, ldq, k, ldk, vt, vt_pitch, out, ldo, batch, heads, n_q, n_kv, head_dim, q_per_kv, scale,
                        key_norm_max, (hipStream_t)stream);
}

namespace {
int attention_impl(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out, int ldo,
                   int batch, int heads, int n_q, int n_kv, int head_dim, int q_per_kv, float scale,
                   const float* key_norm_max, hipStream_t stream) {
  VX_REQUIRE(q && k && vt && out, "vx_attention: null pointer");
  VX_REQUIRE(batch > 0 && heads > 0 && n_q > 0 && n_kv > 0 && q_per_kv > 0 && (batch % q_per_kv) == 0,
             "vx_attention: bad sizes");
  VX_REQUIRE((head_dim % 8) == 0 && (ldq % 8) == 0 && (ldk % 8) == 0 && (vt_pitch % 8) == 0 && (ldo % 4) == 0 &&
                 vt_pitch >= n_kv,
             "vx_attention: alignment (head_dim=%d ldq=%d ldk=%d pitch=%d ldo=%d)", head_dim, ldq, ldk, vt_pitch, ldo);
  VX_REQUIRE((long)batch * heads <= 65535, "vx_attention: batch*heads too large for grid.y");
  // scale == 0: the caller folded scale * log2(e) into K - the scores are base-2 logits already (c = 1 exactly)
  AttnParams p{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)vt, vt_pitch, (bf16_t*)out, ldo,
               batch, heads, n_q, n_kv, head_dim, q_per_kv, scale == 0.f ? 1.0f : scale * 1.4426950408889634f,
               key_norm_max};
  VX_REQUIRE(scale >= 0.f, "vx_attention: scale must be positive (or 0: K carries scale * log2 e)");
  const int d = head_dim;
  static int v1 = -1;
  if (v1 < 0) v1 = getenv("VX_ATTN_V1") != nullptr;
  // attn2: the software-pipelined kernel; instantiated for the head dims whose register budget fits two S tiles
  // (d = 40: the 64x64 level, 85 % of the attention time).  Other head dims keep the plain kernel.
  // With a key-norm table the bounded-softmax body runs (exact fallback inside the kernel); other head dims ignore it.
  if (!v1 && d == 40 && key_norm_max && vx_attn3_variant() > 0 && ((size_t)n_kv * ldk * 2) < (1ull << 31) &&
      ((size_t)d * vt_pitch * 2) < (1ull << 31))
    return vx_attn3_launch(q, ldq, k, ldk, vt, vt_pitch, out, ldo, batch, heads, n_q, n_kv, q_per_kv, p.c, key_norm_max,
                           stream);
  if (!v1 && d > 32 && d <= 48 && (d % 16) != 0)
    return key_norm_max ? launch_attn2<2, 3, 2, true, true>(p, stream) : launch_attn2<2, 3, 2, true, false>(p, stream);
  if (d <= 32) return launch_attn<1, 2, 4, true>(p, stream);
  if (d <= 48) return launch_attn<2, 3, 2, true>(p, stream);
  if (d <= 64) return launch_attn<2, 4, 2, true>(p, stream);
  if (d <= 80) return launch_attn<3, 5, 2, true>(p, stream);
  if (d <= 96) return launch_attn<3, 6, 2, true>(p, stream);
  if (d <= 128) return launch_attn<4, 8, 2, true>(p, stream);
  if (d <= 160) return launch_attn<5, 10, 2, false>(p, stream);
  if (d == 512) return launch_attn<16, 32, 1, false, 1>(p, stream);
  vx_set_error("vx_attention: unsupported head_dim %d", d);
  return VX_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" int vx_temporal_attention(const void* qkv, int ldqkv, void* out, int ldo, int b, int f, int hw, int heads,
                                     int head_dim, float scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(qkv && out, "vx_temporal_attention: null pointer");
  VX_REQUIRE(f >= 1 && f <= 32, "vx_temporal_attention: f=%d outside [1,32] (PositionalEncoding max_len)", f);
  VX_REQUIRE((head_dim % 8) == 0 && (ldqkv % 8) == 0 && (ldo % 4) == 0, "vx_temporal_attention: alignment");
  TemporalParams p{(const bf16_t*)qkv, ldqkv, (bf16_t*)out, ldo, b, f, hw, heads, head_dim,
                   scale * 1.4426950408889634f};
  const int d = head_dim;
  if (d <= 32) return launch_temporal<1, 2>(p, stream);
  if (d <= 64) return launch_temporal<2, 4>(p, stream);
  if (d <= 96) return launch_temporal<3, 6>(p, stream);
  if (d <= 128) return launch_temporal<4, 8>(p, stream);
  if (d <= 160) return launch_temporal<5, 10>(p, stream);
  vx_set_error("vx_temporal_attention: unsupported head_dim %d", d);
  return VX_ERR_UNSUPPORTED;
}

extern "C" int vx_small_kv_attention(const void* q, int ldq, const void* kv, int ldkv, int v_off, void* out, int ldo,
                                     int batch, int n_q, int n_kv, int heads, int head_dim, float scale,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(q && kv && out, "vx_small_kv_attention: null pointer");
  VX_REQUIRE(n_kv >= 1 && n_kv <= SKV_MAX, "vx_small_kv_attention: n_kv=%d outside [1,%d]", n_kv, SKV_MAX);
  VX_REQUIRE((head_dim % 8) == 0 && (ldq % 8) == 0 && (ldkv % 8) == 0 && (v_off % 8) == 0 && (ldo % 8) == 0,
             "vx_small_kv_attention: alignment");
  const int C = heads * head_dim;
  size_t smem = (size_t)n_kv * 2 * C * sizeof(bf16_t);
  VX_REQUIRE(smem <= 64 * 1024, "vx_small_kv_attention: K/V tile %zu B exceeds LDS budget", smem);
  dim3 grid(ceil_div((long)n_q * heads, 256), batch);
  hipLaunchKernelGGL(small_kv_attn_kernel, grid, dim3(256), smem, stream, (const bf16_t*)q, ldq, (const bf16_t*)kv,
                     ldkv, v_off, (bf16_t*)out, ldo, n_q, n_kv, heads, head_dim, scale * 1.4426950408889634f);
  return vx_check_launch("vx_small_kv_attention");
}
